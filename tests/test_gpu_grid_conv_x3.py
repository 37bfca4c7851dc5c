"""Kernel-level parity of the f32x grid-domain convolution (csrc/kernels_conv2d_x3.hip: the 2-D ResNet trunk's layers in the
parity-grade precision mode - f32 rows, three 16-bit matrix instructions per product) against a float64 numpy restatement of
conv3x3 / conv1x1 + bias + ReLU over a [F, T] map (reference libs/nnet/resnet.py:12-20; zero padding = 1 in both axes), every
tile geometry of the kernel, through the layer-program builder of the C ABI (libs/amd/engine.py -> asv_net_*)."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _conv_rows(x, w, b, pos, relu):
    """x [T, F, Cin] float64, w [Cout, Cin, len(pos)], positions (dt, df) -> [T, F, Cout] with zero padding"""
    T, F, _ = x.shape
    y = np.zeros((T, F, w.shape[0]))
    for k, (dt, df) in enumerate(pos):
        src = np.zeros_like(x)
        t0, t1 = max(0, -dt), min(T, T - dt)
        f0, f1 = max(0, -df), min(F, F - df)
        if t0 < t1 and f0 < f1:
            src[t0:t1, f0:f1] = x[t0 + dt:t1 + dt, f0 + df:f1 + df]
        y += src @ w[:, :, k].T
    y += b
    return np.maximum(y, 0.0) if relu else y


POS9 = [(dt, df) for dt in (-1, 0, 1) for df in (-1, 0, 1)]
POS4 = [(-1, -1), (-1, 0), (0, -1), (0, 0)]                # the backward taps of the space-to-depth form
POS1 = [(0, 0)]


def _dense(w, pos, pitch):
    """[Cout, Cin, len(pos)] -> (taps, left, dense [Cout, Cin, span]) in row offsets dt * pitch + df"""
    offs = [dt * pitch + df for dt, df in pos]
    order = np.argsort(offs)
    taps = [offs[i] for i in order]
    left = taps[0]
    dense = np.zeros((w.shape[0], w.shape[1], taps[-1] - left + 1), dtype=np.float32)
    for i in order:
        dense[:, :, offs[i] - left] = w[:, :, i]
    return taps, left, dense


# (feature bins F, channels of the head layer, [(cout, positions, relu)]): which geometry of the kernel each layer lands on
STACKS = {
    "c32_9tap": (80, 32, [(32, POS9, True), (32, POS9, False)]),                                  # Q32 (one chunk, 256-row tiles)
    "c32_odd_bins": (61, 32, [(32, POS9, True)]),
    "c64": (40, 32, [(64, POS1, True), (64, POS9, True), (64, POS4, False)]),                     # Q64P (32 -> 64), Q64, Q64 with backward taps
    "c64_s2d_form": (40, 128, [(64, POS4, True), (64, POS9, False)]),                             # 128 -> 64 over 4 chunks
    "c128": (20, 64, [(128, POS1, True), (128, POS9, True), (128, POS4, True), (128, POS9, False)]),   # Q128P, Q128
    "c256": (10, 128, [(256, POS1, True), (256, POS9, True), (256, POS9, False)]),                # Q256P, Q256
    "c512_two_n_tiles": (10, 64, [(512, POS1, True), (256, POS1, False)]),                        # n tiles of 256 channels; K = 512
    "k1152_im2col_form": (10, 1152, [(256, POS1, True)]),                                         # the im2col'd 128 -> 256 stride-2 layer's GEMM
}


def _build(name, precision, flags=None):
    from libs.amd import engine, ir
    F, c_head, layers = STACKS[name]
    r = np.random.RandomState(len(name) * 131 + F)
    g = ir.Graph(F)
    x = g.grid_input()
    pitch = g.grid_spec(x.tid)[3]
    plan = []
    w0 = (r.randn(c_head, 1, 9) / 3.0).astype(np.float32)
    b0 = (0.2 * r.randn(c_head)).astype(np.float32)
    taps, left, dense = _dense(w0, POS9, pitch)
    v = g.tdnn(x, dense, b0, taps, left, act1="relu")
    plan.append((w0, b0, POS9, True))
    cin = c_head
    for cout, pos, relu in layers:
        w = (r.randn(cout, cin, len(pos)) / np.sqrt(cin * len(pos))).astype(np.float32)
        b = (0.1 * r.randn(cout)).astype(np.float32)
        taps, left, dense = _dense(w, pos, pitch)
        v = g.tdnn(v, dense, b, taps, left, act1="relu" if relu else None)
        plan.append((w, b, pos, relu))
        cin = cout
    g.output = g.pool(v, stddev=True, per_bin=True)
    return engine.Engine(g, device_index=0, precision=precision, flags=flags), plan, F


def _oracle(plan, F, mats):
    out = []
    for m in mats:
        x = np.asarray(m, dtype=np.float64)[:, :, None]                      # [T, F, 1]
        for w, b, pos, relu in plan:
            x = _conv_rows(x, w.astype(np.float64), b.astype(np.float64), pos, relu)
        # per-bin statistics pooling in the device's column order: bin f -> [mean (C) | std (C)]
        mean = x.mean(axis=0)                                                 # [F, C]
        std = np.sqrt(np.maximum(x.var(axis=0), 1e-10))
        out.append(np.concatenate([mean, std], axis=1).reshape(-1))
    return np.stack(out)


def _rel(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - b).max() / np.abs(b).max())


@pytest.mark.parametrize("name", sorted(STACKS))
def test_f32x_grid_convolution_vs_numpy_oracle(name):
    from libs.amd import capi
    F = STACKS[name][0]
    r = np.random.RandomState(7)
    lens = [37, 1, 64, 5, 130] if F * 130 < 6000 else [23, 1, 40]
    mats = [r.randn(T, F).astype(np.float32) for T in lens]
    eng, plan, _ = _build(name, "f32x")
    got = eng.extract_batch(mats).numpy()
    want = _oracle(plan, F, mats)
    assert got.shape == want.shape
    assert np.isfinite(got).all()
    err = _rel(got, want)
    # the split kernel is f32-grade: what is left is the f32 accumulation order (K up to 2304) and the f32 pooling
    assert err < 5e-6, (name, err)
    # the same program on the exact f32-input matrix instruction (the generic tile: ASV_FLAG_SMALL_TILES keeps the new kernel out)
    slow, _, _ = _build(name, "f32x", flags=capi.FLAG_SMALL_TILES)
    ref = slow.extract_batch(mats).numpy()
    assert _rel(ref, want) < 5e-6
    assert not np.array_equal(got, ref), "the f32x grid kernel did not run"
    # the bf16-split variant of the mode (16 significant bits per operand, the full f32 exponent range)
    engb, _, _ = _build(name, "f32x-bf16")
    errb = _rel(engb.extract_batch(mats).numpy(), want)
    assert errb < 2e-4, (name, errb)
    # batch invariance, bit for bit: an utterance alone equals the same utterance inside the batch
    alone = eng.extract_batch([mats[0]]).numpy()
    assert np.array_equal(alone[0], got[0])
    for e in (eng, slow, engb):
        e.close()


def test_persistent_sliding_window_kernels_equal_the_one_tile_kernels(tmp_path):
    """Round 4: the 32 -> 32 and 64 -> 64 convolutions of the first two ResNet stages run as persistent sliding-window kernels in the
    f32x mode (grid_conv_x3_pers32_kernel / _pers64_kernel: the window in an LDS ring, each row fetched and split once, all weight
    fragments in registers).  ASV_AMD_X3_PERS = 0: the one-tile kernels; 3: the 32-channel form only - the one-tile kernel's
    (tap, k-group, term) order, every bit of the embeddings the same; 1 (default): both - the 64-channel form splits K over two waves
    and adds the two chunk sums at the end: equal to f32 rounding.  The switch is read once per process unless ASV_AMD_LIVE_TUNE is
    set: ONE subprocess with it walks the three settings.  Small ragged batch (runs of a few tiles, utterance seams, a 9-frame
    utterance), 12 x 600 frames (many tiles per workgroup, the rings wrap many times) and the golden utterances against the
    reference's outputs, IEEE-half and bf16 halves."""
    import os
    import subprocess
    import sys
    import helpers
    code = r'''
import os, sys, numpy as np
sys.path[:0] = [%r, %r, %r]
import helpers
from libs.amd import synth
g, sd, model = helpers.golden_model("resnet34se_c5")
model.cuda()
sets = {"ragged": [synth.synth_feats(t, 80, 8800 + i) for i, t in enumerate([200, 333, 517, 201, 9])],
        "big": [synth.synth_feats(600, 80, 8900 + i) for i in range(12)], "golden": helpers.golden_feats(g)}
out = {}
for mode in ("0", "1", "3"):
    os.environ["ASV_AMD_X3_PERS"] = mode
    for prec in ("f32x", "f32x-bf16"):
        model.amd_precision = prec
        for name, mats in sets.items():
            out["%%s_%%s_%%s" %% (mode, prec, name)] = model.extract_embedding_batch(mats).numpy()
np.savez(sys.argv[1], **out)
''' % (helpers.REPO, os.path.join(helpers.REPO, "asv-subtools_amd", "pytorch"), os.path.join(helpers.REPO, "tests"))
    path = str(tmp_path / "pers.npz")
    r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, ASV_AMD_LIVE_TUNE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    res = dict(np.load(path))
    g, sd = helpers.golden_state_dict("resnet34se_c5")
    for i in range(len(g["utts"])):
        assert helpers.rel_err(res["1_f32x_golden"][i], g["embeddings"][i]) < 1e-4
    for prec, tol in (("f32x", 2e-6), ("f32x-bf16", 1e-5)):       # (bf16 halves carry 16 significant bits: the mode's own error is ~4e-6)
        for name in ("ragged", "big", "golden"):
            one, both, first = (res["%s_%s_%s" % (m, prec, name)] for m in ("0", "1", "3"))
            assert np.isfinite(both).all() and np.isfinite(first).all()
            assert np.array_equal(first, one), (prec, name, float(np.abs(first - one).max()))
            assert helpers.rel_err(both, one) < tol, (prec, name, helpers.rel_err(both, one))
            assert not np.array_equal(both, one), "the 64-channel persistent form did not run for %s %s" % (prec, name)
