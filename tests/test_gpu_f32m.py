""""f32m" (round 6): the f32x precision mode with its two correction products on gfx950's block-scaled 8-bit matrix instruction
(csrc/kernels_tdnn_chainm.hip, kernels_tdnn_x3m.hip; ASV_FLAG_X3_MX8).  Against the reference's own outputs (tests/golden/*.npz) and the
numpy oracle inside the north star's 1e-4 gate - expected ~1e-5 (tools/emulate_f32m.py) -, with the launch counters proving that the
8-bit kernels are what ran, batch invariance, and the range watch (|x| >= 57344 has no finite e5m2 value: status bit, re-run on the twin)."""

import warnings

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _model(name, prec="f32m"):
    g, sd, model = helpers.golden_model(name)
    model.cuda()
    model.amd_precision = prec
    return g, sd, model


@pytest.mark.parametrize("name", ["xvector_c1", "xvector_near_ragged", "xvector_chunked"])
def test_xvector_f32m_vs_reference_golden(name):
    from libs.amd import capi
    L = capi.lib()
    g, sd, model = _model(name)
    n0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_CHAINM)
    got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_CHAINM) > n0, "the chain did not run on the 8-bit correction kernel"
    worst = 0.0
    for i, (T, _) in enumerate(g["utts"]):
        err = rel_err(got[i], g["embeddings"][i])
        worst = max(worst, err)
        assert err < 1e-4, "%s: utterance of %d frames: %.3g" % (name, T, err)
    print("[f32m] %s: worst embedding error vs the reference %.3g" % (name, worst))


def test_f32m_full_size_batch_against_exact_f32_and_f32x():
    """configs[1]'s batch (256 x 200 frames, 80-dim) in f32m against the exact-f32 extraction of the same engine family: inside the gate
    with margin, and every utterance equal to its extraction in a different batch (an utterance's embedding may not depend on its
    neighbours: same tile arithmetic whatever the batch)."""
    from libs.amd import synth
    model = helpers.build_model("xvector.py", "Xvector(80,10,training=False)")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    import torch
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 0).items()})
    model.cuda()
    mats = [synth.synth_feats(200, 80, 3000 + i) for i in range(256)]
    out = {}
    for prec in ("f32", "f32x", "f32m"):
        model.amd_precision = prec
        out[prec] = model.extract_embedding_batch(mats).numpy()
    e_x, e_m = rel_err(out["f32x"], out["f32"]), rel_err(out["f32m"], out["f32"])
    print("[f32m] 256 x 200: f32x %.3g, f32m %.3g of the exact-f32 extraction" % (e_x, e_m))
    assert e_m < 5e-5 and e_x < 1e-5
    model.amd_precision = "f32m"
    eng = model._amd_engine()
    sub = eng._extract_batch([mats[i] for i in (7, 200, 31)]).numpy()
    for j, i in enumerate((7, 200, 31)):
        # (the rows' products are the same whatever the batch; the f32 merge of the per-tile pooled moments moves with the tile boundaries)
        assert rel_err(sub[j], out["f32m"][i]) < 2e-5, (i, rel_err(sub[j], out["f32m"][i]))


def test_f32m_range_watch_and_rerun():
    """An utterance scaled by 1e5 drives activations past 57344: the kernels raise the status bit, Engine.extract_batch and the scripts'
    DeviceSets re-run the batch on the bf16-halves twin, and every utterance - the ordinary neighbours first of all - is within the gate."""
    from libs.amd import capi
    from libs.amd.pipeline import DeviceSets
    from oracle import np_oracle as O
    g, sd, model = _model("xvector_near_ragged")
    mats = [m.copy() for m in helpers.golden_feats(g)][:6]
    mats[2] = (mats[2] * 1.0e5).astype(np.float32)
    want = np.stack([O.extract_embedding(lambda c: O.xvector_embed(c, sd, "near"), m) for m in mats])
    eng = model._amd_engine()
    assert eng._range_fallback_applies()
    eng.status()
    eng._extract_batch(mats)
    assert eng.status() & capi.STATUS_HALF_RANGE
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = eng.extract_batch(mats).numpy()
    assert any("f32x-bf16" in str(x.message) for x in w)
    for i in range(len(mats)):
        assert rel_err(got[i], want[i]) < 1e-4, i
    rows = sum(m.shape[0] for m in mats)
    offs = np.concatenate([[0], np.cumsum([m.shape[0] for m in mats])]).astype(np.int32)
    sets = DeviceSets(model, rows + 8, 16, mats[0].shape[1], 10000, n_sets=2, results="host")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        sets.host_buffer(0)[:rows] = np.concatenate(mats)
        sets.submit(0, offs, rows)
        got = sets.finish(0).copy()
    assert sets.range_reruns == 1
    for i in range(len(mats)):
        assert rel_err(got[i], want[i]) < 1e-4, i
    clean = [m for i, m in enumerate(mats) if i != 2]
    eng.status()
    got = eng._extract_batch(clean).numpy()
    assert eng.status() == 0
    for j, i in enumerate([0, 1, 3, 4, 5]):
        assert rel_err(got[j], want[i]) < 1e-4, i


X3M_CASES = [
    # (cin, cout, context, utterance lengths, activation, affine_first)
    (512, 512, [-2, 0, 2], [200, 200, 64, 300], "relu", False),           # the x-vector's tdnn2
    (80, 512, [-2, -1, 0, 1, 2], [200, 100, 7], "relu", False),           # tdnn1: 80 channels = two whole 32-channel groups + half of a third
    (512, 1500, [0], [200, 100], "relu", False),                          # 1-tap: a new chunk every pair; 1500 of 1536 channels
    (1024, 1024, [0], [300, 200], None, False),
    (256, 512, [-2, -1, 0, 1, 2], [1, 511, 3], "relu", False),
    (96, 200, [-4, -3, 0, 1, 3, 4], [5, 600], None, False),               # taps up to the halo's edge, 200 of 256 channels
    (32, 256, [0], [129], "relu", False),                                 # ONE pair
    (160, 384, [-1, 0, 1, 2], [40, 41, 300], "tanh", False),              # the generic epilogue
    (192, 320, [-1, 0, 1], [300, 5], "relu", True),                       # bn-relu order
]


@pytest.mark.parametrize("case", X3M_CASES, ids=lambda c: "%dx%d_ctx%s_%s%s" % (c[0], c[1], "_".join(map(str, c[2])), c[4], "_bnfirst" if c[5] else ""))
def test_x3m_kernel_vs_oracle(case, monkeypatch):
    """kernels_tdnn_x3m.hip through the C ABI, forced onto small batches (ASV_AMD_X3M=2): TdnnAffine + activation + eval BN (components.py:107-149,
    365-386, 418-431) against the f64 numpy oracle; the launch counter proves the 8-bit kernel took the layer, and the three-product kernel
    (ASV_AMD_X3M=0) on the same layer is the yardstick: f32m within 3e-5 of the oracle (1e-4 behind a tanh)."""
    import test_gpu_kernels as K
    from libs.amd import capi
    L = capi.lib()
    cin, cout, ctx, lens, act, affine_first = case
    r = np.random.RandomState(cin * 11 + cout)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = r.randn(int(offsets[-1]), cin).astype(np.float32)
    left, right = min(0, ctx[0]), max(0, ctx[-1])
    w = (r.randn(cout, cin, right - left + 1) / np.sqrt(cin * len(ctx))).astype(np.float32)
    b = (0.1 * r.randn(cout)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = (0.2 * r.randn(cout)).astype(np.float32)
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    monkeypatch.setenv("ASV_AMD_P8X", "0")
    monkeypatch.setenv("ASV_AMD_X3M", "2")
    n0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M)
    got = K._tdnn_forward(x, offsets, w, b, ctx, act, scale, shift, affine_first, capi.PREC_F32X, capi.FLAG_X3_MX8)
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M) == n0 + 1, "the 8-bit correction kernel did not take this layer"
    monkeypatch.setenv("ASV_AMD_X3M", "0")
    ref = K._tdnn_forward(x, offsets, w, b, ctx, act, scale, shift, affine_first, capi.PREC_F32X, capi.FLAG_X3_MX8)
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M) == n0 + 1
    want = K._oracle_layer(x, offsets, w, b, ctx, act, scale, shift, affine_first)
    e_m, e_x = rel_err(got, want), rel_err(ref, want)
    print("[x3m] %s: f32m %.3g, three half products %.3g of the oracle" % (case[:3], e_m, e_x))
    # measured on the device: 1.5 - 1.9e-5 for every plain layer (0.5 - 0.9e-6 with three half products); tanh compresses max |y| to ~1.5 and
    # keeps the pre-activation error near 0: 5.3e-5
    assert np.isfinite(got).all() and e_m < (1e-4 if act == "tanh" else 3e-5) and e_x < 3e-6, (e_m, e_x)


def test_x3m_is_the_kernel_of_the_xvector_wide_layers(monkeypatch):
    """Production dispatch in the f32m form at configs[1]'s batch: tdnn1 and tdnn2 go to kernels_tdnn_x3m.hip, the chain to
    kernels_tdnn_chainm.hip; in the plain f32x mode neither runs."""
    import torch
    from libs.amd import capi, synth
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    monkeypatch.setenv("ASV_AMD_X3M", "1")
    L = capi.lib()
    model = helpers.build_model("xvector.py", "Xvector(80,10,training=False)")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 0).items()})
    model.cuda()
    mats = [synth.synth_feats(200, 80, 3000 + i) for i in range(256)]
    for prec, want in (("f32m", (2, 1)), ("f32x", (0, 0))):
        model.amd_precision = prec
        a, b = L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M), L.asv_kernel_launch_count(capi.KERNEL_TDNN_CHAINM)
        model._amd_engine()._extract_batch(mats)
        assert (L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M) - a, L.asv_kernel_launch_count(capi.KERNEL_TDNN_CHAINM) - b) == want, prec


def _synth_xvector():
    import torch
    from libs.amd import synth
    model = helpers.build_model("xvector.py", "Xvector(80,10,training=False)")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 0).items()})
    model.cuda()
    model.amd_precision = "f32m"
    return model


@pytest.mark.parametrize("lens", [[200] * 256, [200] * 255 + [37], list(range(60, 316))], ids=["256x200", "short_last", "ragged_60_315"])
def test_image_rows_between_f32m_layers_change_no_bit(lens, monkeypatch):
    """The hand-over of activations as IMAGES (the [hi halves | x_lo8 | x_hi8] split made once in the producing layer's epilogue instead of
    per workgroup and chunk in the readers: kernels_tdnn_x3m.hip, TdnnKernelParams::y_image / x_image, run_ops' reader_takes_image) is the
    same arithmetic in another place: tdnn1 -> tdnn2 -> chain of configs[1]'s x-vector with it (two image launches per pass, counted) and
    without it (ASV_AMD_X3M_IMAGE=0) must agree to the last bit, on whole and ragged batches."""
    from libs.amd import capi, synth
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    L = capi.lib()
    model = _synth_xvector()
    mats = [synth.synth_feats(T, 80, 5000 + i) for i, T in enumerate(lens)]
    eng = model._amd_engine()
    monkeypatch.setenv("ASV_AMD_X3M_IMAGE", "1")
    n0, m0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE), L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M)
    with_images = eng._extract_batch(mats).numpy()
    assert (L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE) - n0, L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M) - m0) == (2, 2)
    monkeypatch.setenv("ASV_AMD_X3M_IMAGE", "0")
    n1 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE)
    without = eng._extract_batch(mats).numpy()
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE) == n1
    assert np.isfinite(with_images).all() and np.array_equal(with_images, without), float(np.abs(with_images - without).max())


def test_image_rows_through_the_unchained_layers(monkeypatch):
    """With the chain kernel off (ASV_AMD_NO_CHAIN=1: one launch per layer) tdnn1 .. tdnn4 run on the 8-bit wide-layer kernel at configs[1]'s
    batch (tdnn5, with the pooling fused into its epilogue, stays on the three-product kernel and reads f32 rows): three hand-overs as
    images, and the embeddings equal those without them bit for bit and the chained extraction to the f32 order of the pooled moments."""
    from libs.amd import capi, synth
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    L = capi.lib()
    mats = [synth.synth_feats(200, 80, 9000 + i) for i in range(256)]
    chained = _synth_xvector()._amd_engine()._extract_batch(mats).numpy()
    monkeypatch.setenv("ASV_AMD_NO_CHAIN", "1")
    eng = _synth_xvector()._amd_engine()
    res = {}
    for img in ("1", "0"):
        monkeypatch.setenv("ASV_AMD_X3M_IMAGE", img)
        n0, m0, c0 = (L.asv_kernel_launch_count(k) for k in (capi.KERNEL_TDNN_X3M_IMAGE, capi.KERNEL_TDNN_X3M, capi.KERNEL_TDNN_CHAINM))
        res[img] = eng._extract_batch(mats).numpy()
        got = tuple(L.asv_kernel_launch_count(k) - b for k, b in ((capi.KERNEL_TDNN_X3M_IMAGE, n0), (capi.KERNEL_TDNN_X3M, m0), (capi.KERNEL_TDNN_CHAINM, c0)))
        assert got == ((3, 4, 0) if img == "1" else (0, 4, 0)), (img, got)
    assert np.isfinite(res["1"]).all() and np.array_equal(res["1"], res["0"])
    assert rel_err(res["1"], chained) < 2e-5, rel_err(res["1"], chained)


def test_range_watch_through_the_image_epilogue_at_full_size():
    """configs[1]'s batch with ONE utterance scaled by 1e5: with image rows the half-range watch of tdnn2's and the chain's inputs sits in the
    producing layers' epilogues (kernels_tdnn_x3m.hip).  The raw pass must raise ASV_STATUS_HALF_RANGE (and has handed rows over as images),
    the guarded extraction re-runs on the bf16-halves twin, and the 255 ordinary utterances equal their extraction without the scaled
    neighbour to the f32 order of the pooled moments."""
    from libs.amd import capi, synth
    L = capi.lib()
    model = _synth_xvector()
    eng = model._amd_engine()
    mats = [synth.synth_feats(200, 80, 11000 + i) for i in range(256)]
    clean = eng.extract_batch(mats).numpy()
    assert eng.status() == 0
    big = list(mats)
    big[100] = (mats[100] * 1.0e5).astype(np.float32)
    eng.status()
    n0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE)
    eng._extract_batch(big)
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE) == n0 + 2
    assert eng.status() & capi.STATUS_HALF_RANGE
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = eng.extract_batch(big).numpy()
    assert any("f32x-bf16" in str(x.message) for x in w)
    assert np.isfinite(got).all()
    others = [i for i in range(256) if i != 100]
    # (the twin's products are three bf16-halves terms: 1e-5-grade against the f32m pass, far inside the gate)
    assert rel_err(got[others], clean[others]) < 5e-5, rel_err(got[others], clean[others])


def test_image_handover_decisions_over_random_batches(monkeypatch):
    """The producer decides the hand-over by PREDICTING its reader's kernel (run_ops: reader_takes_image); a reader that finds images it cannot
    read fails the call.  40 seeded batches across the sizes where the selection rules flip (the 8-bit kernel from 384 tiles of 128 x 256 on,
    the 8-phase kernel's whole-rounds rule, short / long / tiny utterances, chained and un-chained): no call may fail, and with and without the
    hand-over the embeddings are the same bits."""
    from libs.amd import capi, synth
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    L = capi.lib()
    rng = np.random.default_rng(20260930)
    engines = {"chained": _synth_xvector()._amd_engine()}
    monkeypatch.setenv("ASV_AMD_NO_CHAIN", "1")
    engines["unchained"] = _synth_xvector()._amd_engine()
    monkeypatch.delenv("ASV_AMD_NO_CHAIN")
    pool = {}
    handed = 0
    for trial in range(40):
        n = int(rng.choice([40, 90, 120, 150, 200, 256, 300, 400, 520]))
        kind = trial % 4
        if kind == 0:
            lens = [200] * n
        elif kind == 1:
            lens = [int(v) for v in rng.integers(100, 400, n)]
        elif kind == 2:
            lens = [int(v) for v in rng.integers(1, 60, n // 2)] + [int(v) for v in rng.integers(300, 900, n // 4)]
        else:
            lens = [int(v) for v in rng.integers(150, 260, n)]
        mats = []
        for T in lens:
            if T not in pool:
                pool[T] = synth.synth_feats(T, 80, 13000 + T)
            mats.append(pool[T])
        eng = engines["unchained" if trial % 5 == 4 else "chained"]
        res = {}
        for img in ("1", "0"):
            monkeypatch.setenv("ASV_AMD_X3M_IMAGE", img)
            n0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE)
            res[img] = eng._extract_batch(mats).numpy()
            if img == "1":
                handed += L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE) - n0
        assert np.isfinite(res["1"]).all() and np.array_equal(res["1"], res["0"]), (trial, n, kind)
    assert handed >= 20, handed          # the sweep did reach the hand-over (most batches from ~190 utterances on)
    print("[f32m] 40 random batches: %d hand-overs as images, equal bits with and without" % handed)


def test_image_rows_on_the_small_goldens_and_tiny_utterances(monkeypatch):
    """Forced onto the golden batches (ASV_AMD_X3M=2: the 8-bit kernel from two tiles on): the x-vector goldens inside the gate with image
    rows between its layers; and small batches - one of them a crowd of 5-frame utterances beside long ones - with and without images:
    equal bits."""
    from libs.amd import capi, synth
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    monkeypatch.setenv("ASV_AMD_X3M", "2")
    L = capi.lib()
    for name in ("xvector_c1", "xvector_near_ragged"):
        g, sd, model = _model(name)
        n0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE)
        got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
        assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M_IMAGE) > n0, name
        for i, (T, _) in enumerate(g["utts"]):
            assert rel_err(got[i], g["embeddings"][i]) < 1e-4, (name, T)
    model = _synth_xvector()
    eng = model._amd_engine()
    for lens in ([200] * 8, [5] * 40 + [200] * 4):
        mats = [synth.synth_feats(T, 80, 7000 + i) for i, T in enumerate(lens)]
        res = {}
        for img in ("1", "0"):
            monkeypatch.setenv("ASV_AMD_X3M_IMAGE", img)
            res[img] = eng._extract_batch(mats).numpy()
        assert np.isfinite(res["1"]).all() and np.array_equal(res["1"], res["0"]), (lens[:2], float(np.abs(res["1"] - res["0"]).max()))


@pytest.mark.parametrize("name", ["ecapa_c3", "ecapa_launcher", "ecapa_c512_near_affine", "resnet34se_c5"])
def test_other_models_in_the_f32m_form_vs_reference_golden(name):
    """ECAPA-TDNN's wide layers (1-tap C -> C, 3C -> 1536, the 5-tap input layer) run on kernels_tdnn_x3m.hip in this form (from the production
    tile count on: forced here onto the small golden batches with ASV_AMD_X3M=2), its 128-channel branches and the ResNet's 2-D convolutions stay on their three-product kernels: the mode must
    be inside the gate on every model it can be selected for."""
    import os
    from libs.amd import capi
    L = capi.lib()
    os.environ["ASV_AMD_LIVE_TUNE"] = "1"
    os.environ["ASV_AMD_X3M"] = "2"
    try:
        g, sd, model = _model(name)
        n0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M)
        got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
        ran = L.asv_kernel_launch_count(capi.KERNEL_TDNN_X3M) - n0
    finally:
        os.environ.pop("ASV_AMD_X3M", None)
        os.environ.pop("ASV_AMD_LIVE_TUNE", None)
    worst = max(rel_err(got[i], g["embeddings"][i]) for i in range(len(got)))
    print("[f32m] %s: worst embedding error vs the reference %.3g (%d launches of the 8-bit kernel)" % (name, worst, ran))
    assert worst < 1e-4 and np.isfinite(got).all()
    assert ran > 0 or name.startswith("resnet")
