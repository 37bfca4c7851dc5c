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
        assert rel_err(sub[j], out["f32m"][i]) < 2e-6, (i, rel_err(sub[j], out["f32m"][i]))


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
