"""End-to-end gates of BASELINE.json's north star on synthetic stand-ins for configs C4 / C5:
variable-length utterances -> sharded extraction path -> cosine scoring -> EER, with
|EER(new) - EER(reference-equivalent embeddings)| < 0.01 % absolute on the same trials.

The "reference-equivalent" embeddings are the numpy oracle's (pinned to the reference itself by
tests/golden); trained checkpoints / VoxCeleb are not available, so speakers are planted in the features."""

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _planted_utterances(n_spk, per_spk, dim, t_lo, t_hi, seed, noise=0.8):
    from libs.amd import synth
    r = np.random.RandomState(seed)
    mats, labels = [], []
    for s in range(n_spk):
        base = synth.synth_feats(t_hi, dim, 60000 + 97 * seed + s)
        for u in range(per_spk):
            T = int(r.randint(t_lo, t_hi + 1))
            mats.append((base[:T] + noise * r.randn(T, dim)).astype(np.float32))
            labels.append(s)
    return mats, np.asarray(labels)


def _eer_pipeline(emb, labels, seed):
    from libs.amd import scoring, synth
    ei, ti, tgt = synth.synth_trials(labels, 6000, seed=seed)
    scores = scoring.cosine_trials(emb, emb, ei, ti, submean=scoring.mean_vector(emb))
    eer, thr = scoring.eer(scores, tgt)
    return eer, scores.cpu().numpy()


def test_c4_like_ecapa_sharded_extraction_cosine_eer():
    import torch
    from libs.amd import shard
    from oracle import np_oracle as O
    g, sd, model = helpers.golden_model("ecapa_c512_near_affine")
    model.cuda()
    model.amd_precision = "f32"
    mats, labels = _planted_utterances(24, 5, 80, 200, 500, seed=1)
    lengths = np.array([m.shape[0] for m in mats])
    eng = model._amd_engine()
    dev = torch.device("cuda", 0)

    def extract_batch(batch):
        offs = np.concatenate([[0], np.cumsum([b.shape[0] for b in batch])]).astype(np.int32)
        feats = torch.from_numpy(np.concatenate(batch, axis=0)).to(dev)
        return eng.extract_device(feats, offs)

    emb = shard.extract_sharded(extract_batch, lengths, lambda i: mats[i], max_frames=20000, max_utts=64)
    assert emb.shape == (len(mats), 192)
    want = np.stack([O.extract_embedding(lambda c: O.ecapa_embed(c, sd, "near_affine"), m) for m in mats])
    assert rel_err(emb.cpu().numpy(), want) < 1e-4
    eer_new, s_new = _eer_pipeline(emb, labels, seed=5)
    eer_ref, s_ref = _eer_pipeline(torch.from_numpy(want), labels, seed=5)
    assert 0.0 < eer_ref < 50.0
    assert abs(eer_new - eer_ref) < 0.01, (eer_new, eer_ref)
    assert np.abs(s_new - s_ref).max() < 1e-4


def test_c5_like_resnet_variable_length_cosine_and_plda_eer():
    import torch
    from libs.amd import scoring, synth
    from oracle import np_oracle as O
    g, sd, model = helpers.golden_model("resnet34se_c5")
    model.cuda()
    model.amd_precision = "f32"
    mats, labels = _planted_utterances(10, 4, 80, 200, 320, seed=2)
    emb = model.extract_embedding_batch(mats)
    want = np.stack([O.extract_embedding(lambda c: O.resnet_embed(c, sd, "near", ""), m) for m in mats])
    assert rel_err(emb.numpy(), want) < 1e-4
    eer_new, _ = _eer_pipeline(emb, labels, seed=6)
    eer_ref, _ = _eer_pipeline(torch.from_numpy(want), labels, seed=6)
    assert abs(eer_new - eer_ref) < 0.01, (eer_new, eer_ref)
    # PLDA back-end of config C5 on a planted 256-dim set: device transform + LLR vs the float64 oracle
    from oracle import scoring_oracle as S
    tr_x, tr_l = synth.synth_speaker_embeddings(400, 4, 256, seed=21, within=1.0, between=0.6)
    mean, within, between = scoring.train_plda(tr_x, tr_l, num_iters=3)
    plda = scoring.Plda.from_covariances(mean, within, between)
    ev, ev_l = synth.synth_speaker_embeddings(30, 4, 256, seed=22, within=1.0, between=0.6)
    ei, ti, tgt = synth.synth_trials(ev_l, 3000, seed=23)
    t_dev = plda.transform_vectors(ev)
    llr = plda.llr_trials(t_dev, t_dev, ei, ti).cpu().numpy()
    t_ref = np.stack([S.plda_transform(v.astype(np.float64), plda.mean, plda.transform, plda.psi, 1) for v in ev])
    llr_ref = np.array([S.plda_llr(t_ref[a], 1, t_ref[b], plda.psi) for a, b in zip(ei, ti)])
    e_new, _ = scoring.eer(llr, tgt)
    e_ref, _ = S.compute_eer(llr_ref, tgt)
    assert abs(e_new - 100 * e_ref) < 0.01, (e_new, 100 * e_ref)
