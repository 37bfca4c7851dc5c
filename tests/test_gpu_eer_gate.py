"""The north-star EER gate at the size SURVEY.md 8(d) gives config C4 (VoxCeleb1-O stand-in: 4 708 utterances of
planted speakers, >= 40 000 trials with 50 % targets), for EVERY precision mode the extractor offers: the
parity-grade modes must meet it, the bf16 throughput mode bench.py's headline runs in is measured against it.

    |EER(mode) - EER(reference-equivalent embeddings)| < 0.01 % absolute on the same trials

"Reference-equivalent" = the exact-f32 extraction, which is tied to the numpy oracle (itself pinned to the reference's
outputs) on utterances sampled from inside the run.  With 50 000 trials one flipped trial moves an error rate by 0.004 %,
so the test resolves the 0.01 % gate (the 4 000-trial test it replaces could not)."""

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu

N_SPK, PER_SPK, N_TRIALS = 1177, 4, 50_000            # 4 708 utterances, >= 40 000 trials
GATE = 0.01                                            # percent absolute (BASELINE.json north_star)
# The bf16 throughput mode does NOT meet that gate: measured |delta EER| = 0.024 % abs on this set (at 7.9 % and at 22.8 %
# EER; cosine scores move by up to 4.8e-3 against 4e-5 in f32x mode) - bf16-rounded weights are a slightly different model.
# It is held to a bound just above the measured value so that a regression shows; the parity-grade modes (f32, and f32x -
# the default of the drop-in API) are held to the north-star gate.
BF16_BOUND = 0.05


def _planted(dim, t_lo, t_hi, noise, seed=3):
    from libs.amd import synth
    return synth.synth_planted_utts(N_SPK, PER_SPK, dim, t_lo, t_hi, noise, seed)


def _extract(model, mats, max_frames=130_000):
    """length-sorted batches of <= max_frames frames through the engine (what the extraction script does)"""
    import torch
    eng = model._amd_engine()
    order = np.argsort([-m.shape[0] for m in mats], kind="stable")
    out = np.empty((len(mats), eng.embed_dim), dtype=np.float32)
    i = 0
    while i < len(order):
        j, frames = i, 0
        while j < len(order) and (j == i or frames + mats[order[j]].shape[0] <= max_frames):
            frames += mats[order[j]].shape[0]
            j += 1
        out[order[i:j]] = eng.extract_batch([mats[k] for k in order[i:j]]).numpy()
        i = j
    return out


def _eer(emb, ei, ti, tgt):
    from libs.amd import scoring
    e = scoring.length_normalize(emb, scoring.mean_vector(emb))
    scores = scoring.score_trials(e, e, ei, ti)
    return scoring.eer(scores, tgt)[0], scores.cpu().numpy()


def test_eer_gate_c4_standin_all_precision_modes(capsys):
    import torch
    from libs.amd import synth
    from oracle import np_oracle as O
    model = helpers.build_model("xvector.py", "Xvector(80,10,training=False)")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth.synth_state_dict(shapes, 0)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.cuda()
    mats, labels = _planted(80, 200, 500, noise=0.8)
    assert len(mats) == 4708
    ei, ti, tgt = synth.synth_trials(labels, N_TRIALS, seed=41)
    emb, eer, scores = {}, {}, {}
    # f32x = the default of the drop-in API (IEEE-half hi + lo halves); f32x-bf16 = the round-2 form (bf16 halves); f16 / bf16 = the
    # 16-bit throughput modes; the "-noxlo" / "-nowlo" variants run TWO matrix instructions per product (activations or weights
    # rounded to one half): the measured answer to "is there a mode between one and three instructions that passes the gates"
    modes = ("f32", "f32x", "f32x-bf16", "f16", "bf16", "f32x-f16-noxlo", "f32x-f16-nowlo", "f32x-bf16-noxlo", "f32x-bf16-nowlo")
    for prec in modes:
        model.amd_precision = prec
        emb[prec] = _extract(model, mats)
        eer[prec], scores[prec] = _eer(emb[prec], ei, ti, tgt)
    # the f32 extraction IS the reference-equivalent one: oracle on utterances sampled from inside the run
    pos = [0, 1, 2353, 2354, 4000, 4707]
    want = np.stack([O.extract_embedding(lambda c: O.xvector_embed(c, sd, "far"), mats[i]) for i in pos])
    assert rel_err(emb["f32"][pos], want) < 1e-4
    assert rel_err(emb["f32x"][pos], want) < 1e-4
    assert rel_err(emb["f32x-bf16"][pos], want) < 1e-4
    with capsys.disabled():
        print("\n[eer gate] %d utterances, %d trials, EER f32 %.4f %%" % (len(mats), N_TRIALS, eer["f32"]))
        for prec in modes[1:]:
            print("[eer gate]   %-16s EER delta %+.4f %%   max |score - f32 score| %.2e   embeddings vs f32: max rel %.2e" % (
                prec, eer[prec] - eer["f32"], np.abs(scores[prec] - scores["f32"]).max(), rel_err(emb[prec], emb["f32"])))
    assert 0.5 < eer["f32"] < 30.0, "the planted set should give a non-trivial EER, got %.3f" % eer["f32"]
    assert abs(eer["f32x"] - eer["f32"]) < GATE, (eer["f32x"], eer["f32"])
    assert abs(eer["f32x-bf16"] - eer["f32"]) < GATE, (eer["f32x-bf16"], eer["f32"])
    assert np.abs(scores["f32x"] - scores["f32"]).max() < 2e-5            # the half split is f32-grade: an order below the bf16 split's 4e-5
    assert abs(eer["bf16"] - eer["f32"]) < BF16_BOUND, (eer["bf16"], eer["f32"])
    assert abs(eer["f16"] - eer["f32"]) < BF16_BOUND / 2, (eer["f16"], eer["f32"])
    assert rel_err(emb["f16"], emb["f32"]) < 0.25 * rel_err(emb["bf16"], emb["f32"])        # 3 more significand bits: ~8x closer
    # two matrix instructions per product: the embeddings stay at the one-rounding level of the operand that was not split
    for two in ("f32x-f16-noxlo", "f32x-f16-nowlo"):
        assert rel_err(emb[two], emb["f32"]) > 10 * rel_err(emb["f32x"], emb["f32"]), two


@pytest.mark.parametrize("model", ["xvector", "ecapa", "resnet"])
def test_gates_as_statistics_per_model(model, capsys):
    """one draw (one weight seed, one trial list) cannot settle a 0.01 % gate that is two flipped trials wide.
    tests/gate_table.py: 3 weight seeds x 3 trial lists + one 500 000-trial list per seed, per model, with the model's own scoring
    chain (cosine for the x-vector / ECAPA, PLDA trained on each mode's embeddings for the ResNet).  The parity-grade mode (f32x,
    the default of the drop-in API) must be inside BOTH gates on every draw of every model; the 16-bit throughput modes are
    measured and recorded (gpurun_out/eer_gate_table_<model>.json -> profiles/): they are NOT asserted to pass - the table is
    the answer to "is f16 an EER-grade mode"."""
    import json
    import os
    import gate_table
    g = gate_table.Gates(model)
    tab = g.table(["f32x", "f16", "bf16"], weight_seeds=(0, 1, 2))
    out_dir = os.path.join(helpers.REPO, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "eer_gate_table_%s.json" % model), "w") as f:
        json.dump(tab, f, indent=1)
    with capsys.disabled():
        print("\n[gate table] %s: EER(f32) %s %%" % (model, tab["eer_f32_percent"]))
        for p, m in tab["modes"].items():
            print("[gate table]   %-5s embeddings %.2e (gate %s)   dEER %s | 500k: %s   every draw: %s" % (
                p, m["embedding_max_rel_err"], m["gate_1e-4"], m["eer_delta_percent"], m["eer_delta_percent_500k"], m["eer_gate_every_draw"]))
    x = tab["modes"]["f32x"]
    assert x["gate_1e-4"] and x["eer_gate_every_draw"], x
    assert all(0.3 < e < 45.0 for e in tab["eer_f32_percent"]), tab["eer_f32_percent"]          # non-trivial error rates on every seed
    # the throughput modes: bounded against gross regressions only (measured: DESIGN.md section 5, LABLOG.md "Precision modes")
    assert tab["modes"]["f16"]["embedding_max_rel_err"] < 0.5 * tab["modes"]["bf16"]["embedding_max_rel_err"]
    assert tab["modes"]["bf16"]["worst_abs_eer_delta_percent"] < 3.0
