"""An utterance's embedding must not depend on the batch it is extracted in - for EVERY model, at disparate scale.

The reference extracts every utterance alone (/root/reference/pytorch/libs/nnet/framework.py:33-45); this library packs many utterances
into one row matrix, so every kernel that reduces over an utterance's frames - or over a tile that holds the end of one utterance and
the start of the next - can leak between neighbours.  Round 5 found such a leak in the x-vector chain kernels' fused pooling epilogue
(a stale pivot: relative error 9.8 beside a 1e5 x larger neighbour) that three rounds of like-scale batch-invariance tests had not
seen.  Here the same blind spot is closed for ECAPA-TDNN (attentive_pool_kernel: one-pass sum a x^2 - mean^2; stats_pool_kernel behind
SE_Connect, ecapa_tdnn_xvector.py:97-111; res2_chain_kernel's margin rows, 61-75; eltwise_kernel's per-utterance scale) and for
ResNet34-SE (sum_chunk_kernel / sum_chunk_finish_kernel behind SEBlock_2D, components.py:622-639; the per-bin statistics pooling;
the strided gathers):

  * batches in several orders with one or two utterances scaled by 1e5, every utterance against the numpy oracle (f64 evaluation);
  * a short utterance (<= 40 frames) behind a HUGE filler whose length sweeps the residues of a 32-frame fragment and of the
    128- / 256-row tiles, so that it starts in the last rows of a fragment / tile: "alone" against "beside";
  * through Engine._extract_batch (f32, f32x with bf16 halves, bf16) and through the scripts' path, libs.amd.pipeline.DeviceSets in
    the default f32x mode, where the huge neighbour raises the range status and the whole batch is re-run on the twin.
"""

import warnings

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu

# tolerance of an UNSCALED utterance against the oracle / against its own extraction alone, per mode.  "alone vs beside" differs only
# by the f32 order in which per-tile partial sums are merged (tile boundaries move with the batch), never by the operands.
TOL_ORACLE = {"f32": 1e-4, "f32x-bf16": 1e-4, "f32x": 1e-4, "bf16": 3e-2, "f16": 5e-3}
TOL_ALONE = {"f32": 2e-5, "f32x-bf16": 2e-5, "f32x": 2e-5, "bf16": 2e-3, "f16": 5e-4}
# a 1e5 x scaled utterance against the f64 oracle: its own f32 cancellations (sum a x^2 - mean^2 at 1e10) are the reference's too.
# None = not compared: the ResNet at that input scale is ill-conditioned for 8-bit significands whatever its neighbours are - the BatchNorm
# shifts vanish beside 1e5-scale activations, nothing re-centres them and every convolution carries a large common component (measured
# alone, profiles/r6b_resnet_scale_diag.txt: bf16 1.4e-2 at scale 1, 8e-2 at 10, 0.15 at 100, 0.44 from 1e3 on, bit-identical alone and
# beside; f32 and f32x-bf16 stay at 1e-6 .. 1e-5).  Such an utterance must still be finite and the SAME alone and beside (checked below).
TOL_SCALED = {"ecapa": {"f32": 2e-3, "f32x-bf16": 2e-3, "f32x": 2e-3, "bf16": 6e-2},
              "resnet": {"f32": 2e-3, "f32x-bf16": 2e-3, "f32x": 2e-3, "bf16": None}}

MODELS = {
    "ecapa": ("ecapa_c512_near_affine", lambda O, sd: (lambda c: O.ecapa_embed(c, sd, "near_affine", "relu"))),
    "resnet": ("resnet34se_c5", lambda O, sd: (lambda c: O.resnet_embed(c, sd, "near", "", preact=False))),
}
# (length, seed, scale): two HUGE utterances between ordinary ones; 37 / 23 / 2 frames = shorter than a fragment
BATCH = {
    "ecapa": [(201, 1, 1.0), (200, 2, 1.0e5), (37, 3, 1.0), (129, 4, 1.0), (300, 5, 1.0e5), (64, 6, 1.0), (2, 7, 1.0), (255, 8, 1.0)],
    "resnet": [(88, 1, 1.0), (64, 2, 1.0e5), (23, 3, 1.0), (120, 4, 1.0e5), (40, 5, 1.0), (9, 6, 1.0)],
}
ORDERS = {
    "ecapa": ([0, 1, 2, 3, 4, 5, 6, 7], [7, 6, 5, 4, 3, 2, 1, 0], [4, 2, 1], [1, 6, 4, 2, 0], [3, 4, 5, 1, 2]),
    "resnet": ([0, 1, 2, 3, 4, 5], [5, 4, 3, 2, 1, 0], [3, 2, 1], [1, 5, 3, 4]),
}
_cache = {}


def _setup(kind):
    """(model on the device, matrices, f64-oracle embeddings) - the oracle runs once per model and session."""
    if kind not in _cache:
        from libs.amd import synth
        from oracle import np_oracle as O
        name, make = MODELS[kind]
        g, sd, model = helpers.golden_model(name)
        model.cuda()
        dim = int(g["dim"])
        mats = [(synth.synth_feats(T, dim, 8800 + s) * np.float32(scale)).astype(np.float32) for T, s, scale in BATCH[kind]]
        sd64 = O.cast_state_dict(sd, np.float64)
        fn = make(O, sd64)
        want = np.stack([O.extract_embedding(fn, m, dtype=np.float64) for m in mats])
        assert np.isfinite(want).all()
        _cache[kind] = (model, mats, want, dim)
    return _cache[kind]


def _check(kind, prec, got, order, want):
    for j, i in enumerate(order):
        scaled = BATCH[kind][i][2] != 1.0
        tol = TOL_SCALED[kind][prec] if scaled else TOL_ORACLE[prec]
        assert np.isfinite(got[j]).all(), (kind, prec, order, i)
        if tol is not None:
            assert rel_err(got[j], want[i]) < tol, (kind, prec, order, i, "scaled" if scaled else "neighbour", rel_err(got[j], want[i]))


@pytest.mark.parametrize("prec", ["f32", "f32x-bf16", "bf16"])
@pytest.mark.parametrize("kind", ["ecapa", "resnet"])
def test_every_utterance_matches_the_oracle_beside_huge_neighbours(kind, prec):
    model, mats, want, _ = _setup(kind)
    model.amd_precision = prec
    eng = model._amd_engine()
    for order in ORDERS[kind]:
        got = eng._extract_batch([mats[i] for i in order]).numpy()
        _check(kind, prec, got, order, want)
    # alone against beside, for every utterance of the first order (the scaled ones too: whatever their own conditioning, they may not
    # depend on their neighbours either)
    full = eng._extract_batch(mats).numpy()
    for i in range(len(mats)):
        alone = eng._extract_batch([mats[i]]).numpy()[0]
        assert rel_err(full[i], alone) < TOL_ALONE[prec], (kind, prec, i, rel_err(full[i], alone))


@pytest.mark.parametrize("prec", ["f32x-bf16", "bf16"])
@pytest.mark.parametrize("kind", ["ecapa", "resnet"])
def test_a_short_utterance_at_every_fragment_and_tile_seam(kind, prec):
    """[HUGE filler of L frames | short utterance | HUGE filler]: L sweeps 64 consecutive values (every residue of the 32-frame fragments
    twice, the ends of 128- and 256-row tiles included whatever gap rows the packing inserts), so the short utterance starts in the
    last rows of a fragment / tile for some of them.  Its embedding must be its embedding alone."""
    from libs.amd import synth
    model, _, _, dim = _setup(kind)
    model.amd_precision = prec
    eng = model._amd_engine()
    short_len = {"ecapa": (20, 40, 3), "resnet": (17, 40)}[kind]
    shorts = [synth.synth_feats(T, dim, 8900 + T) for T in short_len]
    alone = [eng._extract_batch([s]).numpy()[0] for s in shorts]
    tail = (synth.synth_feats(77, dim, 8950) * np.float32(1.0e5)).astype(np.float32)
    base = 193 if kind == "ecapa" else 97
    step = 1 if kind == "ecapa" else 1
    worst = 0.0
    for L in range(base, base + 64, step):
        filler = (synth.synth_feats(L, dim, 9000 + L) * np.float32(1.0e5)).astype(np.float32)
        for k, s in enumerate(shorts):
            got = eng._extract_batch([filler, s, tail]).numpy()[1]
            err = rel_err(got, alone[k])
            worst = max(worst, err)
            assert err < TOL_ALONE[prec], (kind, prec, L, s.shape[0], err)
    print("%s %s: worst alone-vs-beside difference over the seam sweep %.3g" % (kind, prec, worst))


@pytest.mark.parametrize("kind", ["ecapa", "resnet"])
def test_default_mode_through_the_scripts_path_reruns_and_keeps_the_neighbours(kind):
    """f32x (the API default) through libs.amd.pipeline.DeviceSets - the extraction scripts' path: the HUGE utterances drive activations
    out of the IEEE-half range of the operand split, the status word behind the batch says so, the batch is re-run on the bf16-halves
    twin, and EVERY utterance of it - the ordinary neighbours first of all - is within the gate of the oracle.  Also
    Engine.extract_batch (the API's guard)."""
    from libs.amd.pipeline import DeviceSets
    model, mats, want, dim = _setup(kind)
    model.amd_precision = "f32x"
    rows = sum(m.shape[0] for m in mats)
    sets = DeviceSets(model, rows + 8, 16, dim, 10000, n_sets=2, results="host")
    assert sets.watch
    for order in ORDERS[kind][:3]:
        sub = [mats[i] for i in order]
        n = sum(m.shape[0] for m in sub)
        offs = np.concatenate([[0], np.cumsum([m.shape[0] for m in sub])]).astype(np.int32)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            before = sets.range_reruns
            sets.host_buffer(0)[:n] = np.concatenate(sub)
            sets.submit(0, offs, n)
            got = sets.finish(0).copy()
        assert sets.range_reruns == before + 1 and any("f32x-bf16" in str(x.message) for x in w), (kind, order)
        _check(kind, "f32x", got, order, want)
        # a clean batch straight after the flagged one: no stale status, no re-run, the default mode's own arithmetic
        clean = [i for i in order if BATCH[kind][i][2] == 1.0]
        sub = [mats[i] for i in clean]
        n = sum(m.shape[0] for m in sub)
        offs = np.concatenate([[0], np.cumsum([m.shape[0] for m in sub])]).astype(np.int32)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            sets.host_buffer(1)[:n] = np.concatenate(sub)
            sets.submit(1, offs, n)
            got = sets.finish(1).copy()
        assert sets.range_reruns == before + 1 and not w
        _check(kind, "f32x", got, clean, want)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = model._amd_engine().extract_batch(mats).numpy()
    assert any("f32x-bf16" in str(x.message) for x in w)
    _check(kind, "f32x", got, list(range(len(mats))), want)
