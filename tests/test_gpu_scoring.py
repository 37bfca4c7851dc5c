"""Scoring kernels on the MI355X against the oracle and the reference-generated fixture."""

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _sets(dim=192, seed=3):
    from libs.amd import synth
    x, labels = synth.synth_speaker_embeddings(60, 6, dim, seed=seed, within=1.0, between=0.7)
    ei, ti, tgt = synth.synth_trials(labels, 5000, seed=seed + 1)
    return x, labels, ei, ti, tgt


def test_cosine_trials_and_matrix_vs_oracle():
    from libs.amd import scoring
    from oracle import scoring_oracle as S
    x, labels, ei, ti, tgt = _sets()
    mean = scoring.mean_vector(x).cpu().numpy()
    assert rel_err(mean, S.global_mean(x)) < 1e-5
    got = scoring.cosine_trials(x, x, ei, ti, submean=mean).cpu().numpy()
    xn = S.length_normalize(x, S.global_mean(x))
    want = S.dot_trials(xn, xn, ei, ti)
    assert np.abs(got - want).max() < 2e-6
    mat = scoring.score_matrix(scoring.length_normalize(x, mean), scoring.length_normalize(x, mean)).cpu().numpy()
    assert mat.shape == (360, 360)
    assert np.abs(mat - xn.dot(xn.T)).max() < 2e-6
    assert np.abs(mat[ei, ti] - got).max() < 2e-6                     # trial gather == matrix entries


def test_eer_matches_reference_semantics_and_is_identical_for_oracle_embeddings():
    from libs.amd import scoring
    from oracle import scoring_oracle as S
    x, labels, ei, ti, tgt = _sets()
    scores = scoring.cosine_trials(x, x, ei, ti).cpu().numpy()
    eer_gpu, thr_gpu = scoring.eer(scores, tgt)
    eer_cpu, thr_cpu = S.compute_eer(scores, tgt)
    assert abs(eer_gpu - 100 * eer_cpu) < 1e-4 and abs(thr_gpu - thr_cpu) < 1e-6
    g = np.load(helpers.GOLDEN + "/scoring_plda.npz")                 # fixture from the reference's own EER code
    e, t = scoring.eer(g["llr"].astype(np.float32), g["trials_tgt"])
    assert abs(e - 100 * float(g["eer"])) < 1e-3


def test_eer_with_ties_and_extremes():
    from libs.amd import scoring
    from oracle import scoring_oracle as S
    r = np.random.RandomState(0)
    scores = np.round(r.randn(2000), 1).astype(np.float32)            # many exact ties
    labels = (r.rand(2000) < 0.3).astype(np.int32)
    scores[labels == 1] += 0.5
    e_gpu, t_gpu = scoring.eer(scores, labels)
    e_cpu, t_cpu = S.compute_eer(scores, labels)
    assert abs(e_gpu - 100 * e_cpu) < 1e-4 and abs(t_gpu - t_cpu) < 1e-6


def test_plda_transform_and_llr_vs_reference_fixture():
    from libs.amd import scoring, synth
    g = np.load(helpers.GOLDEN + "/scoring_plda.npz")
    dim = int(g["dim"])
    ev, _ = synth.synth_speaker_embeddings(40, 5, dim, seed=12, within=1.0, between=0.8)
    plda = scoring.Plda(g["mean"], g["transform"], g["psi"])
    tr = plda.transform_vectors(ev)
    assert rel_err(tr.cpu().numpy(), g["transformed"]) < 2e-5
    llr = plda.llr_trials(tr, tr, g["trials_e"], g["trials_t"]).cpu().numpy()
    assert np.abs(llr - g["llr"]).max() < 2e-3 * max(1.0, np.abs(g["llr"]).max() / 10)
    e_new, _ = scoring.eer(llr, g["trials_tgt"])
    assert abs(e_new - 100 * float(g["eer"])) < 0.01                  # |EER delta| < 0.01 % absolute
    # simple length norm: ||y|| == sqrt(dim)
    ys = plda.transform_vectors(ev, simple_length_norm=True).cpu().numpy()
    assert np.allclose(np.linalg.norm(ys, axis=1), np.sqrt(dim), rtol=1e-5)


def test_eer_delta_between_extractor_precisions_is_negligible():
    """north_star gate: EER(new) vs EER(oracle embeddings) on the same trials < 0.01 % abs -
    here between the f32 (parity) and bf16 (throughput) extractors on a planted-speaker set."""
    from libs.amd import scoring, synth
    g, sd, model = helpers.golden_model("xvector_near_ragged")
    model.cuda()
    r = np.random.RandomState(9)
    n_spk, per = 24, 5
    base = [synth.synth_feats(120, 80, 40000 + s) for s in range(n_spk)]
    mats, labels = [], []
    for s in range(n_spk):
        for u in range(per):
            mats.append((base[s] + 0.6 * r.randn(120, 80)).astype(np.float32))
            labels.append(s)
    ei, ti, tgt = synth.synth_trials(np.array(labels), 4000, seed=77)
    eers = {}
    for prec in ("f32", "bf16"):
        model.amd_precision = prec
        emb = model.extract_embedding_batch(mats)
        scores = scoring.cosine_trials(emb, emb, ei, ti, submean=scoring.mean_vector(emb))
        eers[prec], _ = scoring.eer(scores, tgt)
    assert abs(eers["f32"] - eers["bf16"]) < 0.25, eers        # a handful of trials of 4000 may flip at the threshold
