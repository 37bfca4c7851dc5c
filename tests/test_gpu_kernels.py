"""Kernel-level parity on the MI355X: every GEMM code path (f32 MFMA, bf16 MFMA and the
plain-VALU self-check kernels) against the numpy oracle, through the C ABI."""

import ctypes as C

import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _dev(arr):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).cuda()


def _tdnn_forward(x, offsets, w, b, context, act, scale, shift, affine_first, precision, flags):
    import torch
    from libs.amd import capi
    L = capi.lib()
    left = min(0, context[0])
    d = capi.TdnnDesc()
    d.struct_size = C.sizeof(capi.TdnnDesc)
    d.in_ch, d.out_ch = w.shape[1], w.shape[0]
    d.n_taps = len(context)
    for i, t in enumerate(context):
        d.taps[i] = t
    w = np.ascontiguousarray(w, dtype=np.float32)
    d.weight = capi.f32_ptr(w)
    d.w_tot_context, d.w_left_context = w.shape[2], left
    d.bias = capi.f32_ptr(b) if b is not None else None
    d.act1 = capi.ACT_BY_NAME[act]
    d.scale = capi.f32_ptr(scale) if scale is not None else None
    d.shift = capi.f32_ptr(shift) if shift is not None else None
    d.affine_first = int(affine_first)
    d.in2_buf = d.seg_bias_buf = d.seg_scale_buf = d.res_buf = -1
    xd = _dev(x)
    yd = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device="cuda")
    offs = np.ascontiguousarray(offsets, dtype=np.int32)
    capi.check(L.asv_tdnn_forward(C.byref(d), precision, flags, C.c_void_p(xd.data_ptr()), offs.ctypes.data_as(capi.c_int32_p),
                                  len(offs) - 1, C.c_void_p(yd.data_ptr()), None), "asv_tdnn_forward")
    return yd.cpu().numpy()


def _oracle_layer(x, offsets, w, b, context, act, scale, shift, affine_first):
    from oracle import np_oracle as O
    outs = []
    for u in range(len(offsets) - 1):
        seg = x[offsets[u]:offsets[u + 1]].astype(np.float64)
        z = O.tdnn_affine(seg, w.astype(np.float64), None if b is None else b.astype(np.float64), list(context))
        s = 1.0 if scale is None else scale.astype(np.float64)
        t = 0.0 if shift is None else shift.astype(np.float64)
        z = O._act(z * s + t, act) if affine_first else O._act(z, act) * s + t
        outs.append(z)
    return np.concatenate(outs, axis=0)


CASES = [
    # (in_ch, out_ch, context, lengths)
    (80, 512, [-2, -1, 0, 1, 2], [200, 3, 1, 57, 200, 131]),
    (30, 512, [-2, -1, 0, 1, 2], [200, 199, 2]),
    (512, 512, [-2, 0, 2], [200, 200, 64, 300]),
    (512, 512, [-3, 0, 3], [129, 1, 2, 3, 4, 5, 6, 7, 250]),
    (512, 1500, [0], [200, 100]),
    (128, 128, [-4, 0, 4], [300, 9, 8]),
    (96, 200, [0], [5]),
    (1536, 128, [0], [300, 200, 77]),          # bf16: the 128 x 128 geometry of the wide kernel (ECAPA's attention bottleneck)
    (576, 128, [0], [260, 130, 1]),            # (the im2col'd 64 -> 128 convolution of the ResNet trunk)
]


def _mode_args(mode):
    """test mode name -> (precision id, flags) of the C ABI"""
    from libs.amd import capi
    base, _, opt = mode.partition("_")
    prec = {"f32": capi.PREC_F32, "bf16": capi.PREC_BF16, "f16": capi.PREC_F16, "f32x": capi.PREC_F32X, "f32xb": capi.PREC_F32X, "f32x128": capi.PREC_F32X}[base]
    flags = capi.FLAG_REF_KERNELS if opt == "ref" else 0
    if base == "f32xb":
        flags |= capi.FLAG_X3_SPLIT_BF16
    if base == "f32x128":
        flags |= capi.FLAG_X3_TILE128            # small batches take the 64-row geometry of the split kernel by themselves
    return prec, flags


# bf16: 8-bit significand operands and outputs; f16: 11 bits (8x finer); f32xb: bf16 hi + lo operand halves (16 bits, the
# lo * lo term dropped); f32x: IEEE-half hi + lo halves (22 bits): what is left is the f32 accumulation itself
MODE_TOL = {"bf16": 2e-2, "f16": 2.5e-3, "f32xb": 2e-5, "f32x": 3e-6, "f32x128": 3e-6, "f32": 2e-5}


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d_ctx%s" % (c[0], c[1], "_".join(map(str, c[2]))))
@pytest.mark.parametrize("mode", ["f32_mfma", "f32_ref", "bf16_mfma", "bf16_ref", "f16_mfma", "f16_ref", "f32x_mfma", "f32xb_mfma", "f32x128_mfma"])
def test_tdnn_layer_vs_oracle(case, mode):
    cin, cout, ctx, lens = case
    r = np.random.RandomState(cin * 7 + cout)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = r.randn(int(offsets[-1]), cin).astype(np.float32)
    left, right = min(0, ctx[0]), max(0, ctx[-1])
    w = (r.randn(cout, cin, right - left + 1) / np.sqrt(cin * len(ctx))).astype(np.float32)   # dense kernel, garbage in masked taps
    b = (0.1 * r.randn(cout)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = (0.2 * r.randn(cout)).astype(np.float32)
    prec, flags = _mode_args(mode)
    got = _tdnn_forward(x, offsets, w, b, ctx, "relu", scale, shift, False, prec, flags)
    want = _oracle_layer(x, offsets, w, b, ctx, "relu", scale, shift, False)
    err = rel_err(got, want)
    wide = cout >= 192 and cin >= 32                # only the wide layers run on the split kernel; the others are exact f32 in the f32x modes
    tol = MODE_TOL[mode.split("_")[0]] if (wide or not mode.startswith("f32x")) else 2e-5
    assert err < tol, "%s: rel err %g" % (mode, err)


def test_half_split_keeps_small_operands_and_two_products_are_not_enough():
    """The f32x mode's IEEE-half split (kernels_tdnn_x3.hip): (1) activations of magnitude 1e-2 .. 1e-3 have a SUBNORMAL lo half
    - the matrix cores must not flush it (they would leave 2^-12 relative error, the one-rounding level); weights of any
    magnitude are lifted into the normal range by the host's per-layer power of two.  (2) the measurement variants that drop
    one of the three products land at the one-rounding level of the dropped operand, three orders above the full split."""
    from libs.amd import capi
    r = np.random.RandomState(33)
    lens = [200, 131, 64]
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    errs = {}
    for xs, ws in ((1.0, 1.0), (3e-3, 1.0), (1.0, 1e-4), (40.0, 30.0)):
        x = (xs * r.randn(int(offsets[-1]), 512)).astype(np.float32)
        w = (ws * r.randn(512, 512, 5) / 40).astype(np.float32)
        want = _oracle_layer(x, offsets, w, None, [-2, 0, 2], None, None, None, False)
        for name, flags in (("f16x3", 0), ("bf16x3", capi.FLAG_X3_SPLIT_BF16), ("f16 no x_lo", capi.FLAG_X3_NO_XLO), ("f16 no w_lo", capi.FLAG_X3_NO_WLO),
                            ("f16 one product", capi.FLAG_X3_NO_XLO | capi.FLAG_X3_NO_WLO)):
            got = _tdnn_forward(x, offsets, w, None, [-2, 0, 2], None, None, None, False, capi.PREC_F32X, flags)
            errs[(xs, ws, name)] = rel_err(got, want)
    print("\n[x3 split] " + "; ".join("x*%g w*%g %s: %.2e" % (k + (v,)) for k, v in errs.items()))
    for xs, ws in ((1.0, 1.0), (3e-3, 1.0), (1.0, 1e-4), (40.0, 30.0)):
        # activations of magnitude 3e-3: hi keeps 11 bits, lo is a subnormal half with the absolute precision 2^-25 = 1e-5 of such a
        # value (kept, not flushed: a flush would leave 2^-12 = 2.4e-4) - the level of the bf16 split; O(1) activations: f32-grade
        assert errs[(xs, ws, "f16x3")] < (1e-5 if xs < 0.1 else 3e-6), (xs, ws, errs[(xs, ws, "f16x3")])
        assert errs[(xs, ws, "bf16x3")] < 2e-5
        for two in ("f16 no x_lo", "f16 no w_lo"):
            assert 2e-5 < errs[(xs, ws, two)] < 2e-3, (two, errs[(xs, ws, two)])


def test_bf16_mfma_matches_bf16_ref_closely():
    """Same bf16-rounded operands, f32 accumulation: only the summation order differs."""
    from libs.amd import capi
    r = np.random.RandomState(5)
    lens = [200, 77, 130]
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = r.randn(int(offsets[-1]), 512).astype(np.float32)
    w = (r.randn(512, 512, 5) / 40).astype(np.float32)
    a = _tdnn_forward(x, offsets, w, None, [-2, 0, 2], None, None, None, False, capi.PREC_BF16, 0)
    b = _tdnn_forward(x, offsets, w, None, [-2, 0, 2], None, None, None, False, capi.PREC_BF16, capi.FLAG_REF_KERNELS)
    # outputs are bf16-rounded: at most one bf16 ulp apart where the f32 sums straddle a rounding boundary
    assert rel_err(a, b) < 1e-2
    assert np.mean(a == b) > 0.97


def test_bn_relu_order_and_activations():
    from libs.amd import capi
    r = np.random.RandomState(11)
    offsets = np.array([0, 150, 151], dtype=np.int32)
    x = r.randn(151, 64).astype(np.float32)
    w = (r.randn(96, 64, 1) / 8).astype(np.float32)
    b = (0.1 * r.randn(96)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, 96).astype(np.float32)
    shift = (0.2 * r.randn(96)).astype(np.float32)
    for act in ("relu", "tanh", "sigmoid", None):
        for first in (False, True):
            got = _tdnn_forward(x, offsets, w, b, [0], act, scale, shift, first, capi.PREC_F32, 0)
            want = _oracle_layer(x, offsets, w, b, [0], act, scale, shift, first)
            # 256 products of unit-scale operands, each carrying the 2^-17 representation error of a bf16 pair: ~3e-5 at the
        # worst of 160 000 outputs (the embedding-level 1e-4 gate is tests/test_gpu_full_size_parity.py)
        assert rel_err(got, want) < 5e-5, (act, first)


def test_bf16_generic_epilogue_of_the_wide_kernel_tanh_sigmoid():
    """bf16 layers with tanh / sigmoid (or the bn-relu order) and >= 97 output channels run on the wide kernel's generic
    epilogue, whose tanh / sigmoid are built on v_exp_f32 + v_rcp_f32: against the numpy oracle at bf16 tolerance, on the
    128 x 128 geometry (128 channels) and the 128 x 256 one (512 channels), saturating arguments included."""
    from libs.amd import capi
    r = np.random.RandomState(21)
    offsets = np.array([0, 200, 333, 334], dtype=np.int32)
    for cin, cout in ((1024, 128), (512, 512)):
        x = r.randn(334, cin).astype(np.float32)
        w = (r.randn(cout, cin, 1) / np.sqrt(cin)).astype(np.float32)                  # unit-scale pre-activations (bf16 operands: ~3e-3 absolute)
        b = (0.5 * r.randn(cout)).astype(np.float32)
        b[:8] = [15.0, -15.0, 40.0, -40.0, 90.0, -90.0, 200.0, -200.0]                  # saturated channels: exp overflow / underflow inside the fast forms
        scale = r.uniform(0.5, 1.5, cout).astype(np.float32)
        shift = (0.2 * r.randn(cout)).astype(np.float32)
        for act in ("tanh", "sigmoid"):
            for first in (False, True):
                got = _tdnn_forward(x, offsets, w, b, [0], act, scale, shift, first, capi.PREC_BF16, 0)
                want = _oracle_layer(x, offsets, w, b, [0], act, scale, shift, first)
                assert np.isfinite(got).all(), (cin, cout, act, first)
                assert rel_err(got, want) < 2e-2, (cin, cout, act, first, rel_err(got, want))


@pytest.mark.parametrize("lens", [[200, 200, 200], [1, 2, 3, 1000, 17], [513]])
@pytest.mark.parametrize("channels", [1500, 64, 100])
def test_stats_pool_vs_oracle(lens, channels):
    import torch
    from libs.amd import capi
    from oracle import np_oracle as O
    L = capi.lib()
    r = np.random.RandomState(len(lens) + channels)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = (r.randn(int(offsets[-1]), channels) * r.uniform(0.1, 3, channels) + r.randn(channels)).astype(np.float32)
    xd = _dev(x)
    yd = torch.empty((len(lens), 2 * channels), dtype=torch.float32, device="cuda")
    capi.check(L.asv_stats_pool_forward(C.c_void_p(xd.data_ptr()), channels, offsets.ctypes.data_as(capi.c_int32_p), len(lens), 1, 0, 0,
                                        C.c_float(1e-10), C.c_void_p(yd.data_ptr()), None), "asv_stats_pool_forward")
    got = yd.cpu().numpy()
    want = np.stack([O.statistics_pooling(x[offsets[u]:offsets[u + 1]].astype(np.float64)) for u in range(len(lens))])
    assert rel_err(got, want) < 1e-5


@pytest.mark.parametrize("case", [(512, 512, [-2, 0, 2], [200] * 5 + [77, 3]), (80, 512, [-2, -1, 0, 1, 2], [200, 31]),
                                  (512, 1500, [0], [255, 257, 1]), (96, 200, [0], [300])],
                         ids=lambda c: "%dx%d" % (c[0], c[1]))
def test_big_tile_kernel_matches_small_tile_kernel(case):
    """The 256x256 direct-to-LDS kernel and the 128x128 register-staged kernel consume the same
    bf16 operands with f32 accumulation: outputs agree to the last bf16 rounding."""
    from libs.amd import capi
    cin, cout, ctx, lens = case
    r = np.random.RandomState(cin + cout)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = r.randn(int(offsets[-1]), cin).astype(np.float32)
    left, right = min(0, ctx[0]), max(0, ctx[-1])
    w = (r.randn(cout, cin, right - left + 1) / np.sqrt(cin * len(ctx))).astype(np.float32)
    b = (0.1 * r.randn(cout)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = (0.2 * r.randn(cout)).astype(np.float32)
    big = _tdnn_forward(x, offsets, w, b, ctx, "relu", scale, shift, False, capi.PREC_BF16, 0)
    small = _tdnn_forward(x, offsets, w, b, ctx, "relu", scale, shift, False, capi.PREC_BF16, capi.FLAG_SMALL_TILES)
    want = _oracle_layer(x, offsets, w, b, ctx, "relu", scale, shift, False)
    assert rel_err(big, want) < 2e-2 and rel_err(small, want) < 2e-2
    assert rel_err(big, small) < 1e-2
    assert np.mean(big == small) > 0.97


def test_f32x_split_kernel_is_f32_grade_on_hard_inputs():
    """The f32x kernel (kernels_tdnn_x3.hip: w_hi x_hi + w_hi x_lo + w_lo x_hi on the bf16 matrix cores) against the float64
    oracle on inputs that punish a plain bf16 product: a large common offset on every input channel (the information sits in
    the low mantissa bits), wide dynamic range across channels, a 1536-deep contraction, ragged segments with one-frame
    utterances, and the generic epilogue (tanh, BN before the activation)."""
    from libs.amd import capi
    r = np.random.RandomState(21)
    lens = [200, 1, 64, 65, 300, 2]
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    cin, cout, ctx = 512, 512, [-3, 0, 3]
    x = (r.randn(int(offsets[-1]), cin) * np.exp(r.uniform(-3, 3, cin)) + 40.0 * r.randn(cin)).astype(np.float32)
    w = (r.randn(cout, cin, 7) / np.sqrt(3 * cin)).astype(np.float32)
    b = (0.1 * r.randn(cout)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = (0.2 * r.randn(cout)).astype(np.float32)
    want = _oracle_layer(x, offsets, w, b, ctx, None, scale, shift, False)
    got = _tdnn_forward(x, offsets, w, b, ctx, None, scale, shift, False, capi.PREC_F32X, 0)
    exact = _tdnn_forward(x, offsets, w, b, ctx, None, scale, shift, False, capi.PREC_F32, 0)
    plain = _tdnn_forward(x, offsets, w, b, ctx, None, scale, shift, False, capi.PREC_BF16, 0)
    assert rel_err(exact, want) < 2e-6
    assert rel_err(got, want) < 2e-5, rel_err(got, want)
    assert rel_err(plain, want) > 20 * rel_err(got, want)              # the split is what buys the accuracy
    for act, first in (("tanh", True), ("sigmoid", False)):
        x2 = r.randn(int(offsets[-1]), 256).astype(np.float32)
        w2 = (r.randn(256, 256, 1) / 16).astype(np.float32)
        got = _tdnn_forward(x2, offsets, w2, b[:256], [0], act, scale[:256], shift[:256], first, capi.PREC_F32X, 0)
        want = _oracle_layer(x2, offsets, w2, b[:256], [0], act, scale[:256], shift[:256], first)
        # 256 products of unit-scale operands, each carrying the 2^-17 representation error of a bf16 pair: ~3e-5 at the
        # worst of 160 000 outputs (the embedding-level 1e-4 gate is tests/test_gpu_full_size_parity.py)
        assert rel_err(got, want) < 5e-5, (act, first)


def test_im2col_descriptor_zero_initialised_means_no_optional_buffers():
    """asv_im2col_desc_t gained b_buf / seg_scale_buf in round 4 (-1 = none).  A C caller that zero-initialises the
    descriptor (memset + struct_size) passes 0 in both - buffer 0 is the feature matrix, never a grid addend or a per-segment scale -
    and must get the plain gather it got before the fields existed, not 'the addend must be a whole buffer...'."""
    from libs.amd import capi
    lib = capi.lib()
    net = C.c_void_p()
    capi.check(lib.asv_net_create(C.byref(net), 0, capi.PREC_BF16, 0, 80), "asv_net_create")
    try:
        g0 = capi.check(lib.asv_net_define_grid(net, 0, 80, 83), "asv_net_define_grid")
        g1 = capi.check(lib.asv_net_define_grid(net, 1, 40, 43), "asv_net_define_grid")
        a = capi.check(lib.asv_net_new_buffer(net, g0, 32), "asv_net_new_buffer")
        b = capi.check(lib.asv_net_new_buffer(net, g1, 32 * 4), "asv_net_new_buffer")
        d = capi.Im2colDesc()                                    # ctypes structures start zeroed: b_buf = seg_scale_buf = 0
        d.struct_size = C.sizeof(capi.Im2colDesc)
        d.in_buf, d.out_buf, d.channels, d.n_taps, d.stride = a, b, 32, 4, 2
        for i, (dt, df) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            d.dt[i], d.df[i] = dt, df
        assert d.b_buf == 0 and d.seg_scale_buf == 0
        capi.check(lib.asv_net_add_im2col(net, C.byref(d)), "asv_net_add_im2col (zero-initialised optional buffers)")
        d.b_buf = -1
        d.seg_scale_buf = -1
        capi.check(lib.asv_net_add_im2col(net, C.byref(d)), "asv_net_add_im2col (-1)")
    finally:
        lib.asv_net_destroy(net)


P8_CASES = [
    # layers the 8-phase kernel takes (whole 64-channel chunks, an even number of K-tiles, >= 192 output channels, plain epilogue)
    (512, 512, [-2, 0, 2], [200, 200, 64, 300]),
    (512, 512, [-3, 0, 3], [129, 1, 2, 3, 4, 5, 6, 7, 250]),
    (512, 1500, [0], [200, 100]),
    (128, 256, [0], [300, 9, 8]),                  # two K-tiles: prologue + the two closing K-tiles only
    (192, 320, [-1, 1], [77, 256, 1, 255]),        # 3 chunks x 2 taps, a partly filled last channel tile
    (1024, 1024, [0], [300, 200]),
    (64, 200, [-4, 4], [5, 600]),                  # one chunk x two taps at the halo's edge, 200 of 256 channels
    (256, 512, [-2, -1, 0, 1, 2], [1, 511, 3]),    # 4 chunks x 5 taps = 20 K-tiles
    (192, 512, [0], [130, 200]),                   # three K-tiles: an odd count (prologue + one closing pair + the last)
    (64, 256, [0], [300]),                         # ONE K-tile
    (320, 384, [-1, 0, 1], [40, 41, 300]),         # 5 chunks x 3 taps = 15
]


@pytest.mark.parametrize("case", P8_CASES, ids=lambda c: "%dx%d_ctx%s" % (c[0], c[1], "_".join(map(str, c[2]))))
@pytest.mark.parametrize("mode", ["bf16", "f16"])
@pytest.mark.parametrize("act", ["relu", None])
def test_p8_kernel_vs_oracle_and_big3(case, mode, act, monkeypatch):
    """Round 5: kernels_tdnn_p8.hip (256 x 256 tiles, both operands through LDS-DMA, four phases per K-tile, staggered wave rows)
    through the C ABI, forced onto small batches (ASV_AMD_P8=2: from two tiles on; production: from one round of the chip's CUs):
    against the f64 numpy oracle of TdnnAffine + ReLU + eval BN (components.py:107-149, 418-431) AND bit for bit against the
    variant-3 kernel (same products in the same order) - ragged utterances, gap rows, taps up to the halo, partly filled channel
    tiles.  The launch counters prove which kernel ran."""
    from libs.amd import capi
    L = capi.lib()
    cin, cout, ctx, lens = case
    r = np.random.RandomState(cin * 11 + cout)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = r.randn(int(offsets[-1]), cin).astype(np.float32)
    left, right = min(0, ctx[0]), max(0, ctx[-1])
    w = (r.randn(cout, cin, right - left + 1) / np.sqrt(cin * len(ctx))).astype(np.float32)
    b = (0.1 * r.randn(cout)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = (0.2 * r.randn(cout)).astype(np.float32)
    prec, flags = _mode_args(mode + "_mfma")
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    monkeypatch.setenv("ASV_AMD_P8", "2")
    n0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8)
    got = _tdnn_forward(x, offsets, w, b, ctx, act, scale, shift, False, prec, flags)
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8) == n0 + 1, "the 8-phase kernel did not take this layer"
    monkeypatch.setenv("ASV_AMD_P8", "0")
    m0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_BIG3)
    ref = _tdnn_forward(x, offsets, w, b, ctx, act, scale, shift, False, prec, flags)
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_BIG3) == m0 + 1 and L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8) == n0 + 1
    assert np.array_equal(got, ref), "8-phase kernel and variant-3 kernel differ in %d values" % int((got != ref).sum())
    want = _oracle_layer(x, offsets, w, b, ctx, act, scale, shift, False)
    assert rel_err(got, want) < MODE_TOL[mode]


P8X_CASES = [
    # layers the f32x 8-phase kernel takes (whole 32-channel chunks, cin x taps >= 512, >= 192 output channels, plain epilogue)
    (512, 512, [-2, 0, 2], [200, 200, 64, 300]),
    (512, 1500, [0], [200, 100]),
    (1024, 1024, [0], [300, 200]),
    (256, 512, [-2, -1, 0, 1, 2], [1, 511, 3]),    # 8 chunks x 5 taps = 40 K-tiles
    (96, 200, [-4, -3, 0, 1, 3, 4], [5, 600]),     # 3 chunks x 6 taps = 18, taps up to the halo's edge, 200 of 256 channels
    (512, 320, [0], [77, 256, 1, 255]),            # a partly filled second channel tile
    (544, 256, [0], [300]),                        # 17 K-tiles: an odd count
    (160, 384, [-1, 0, 1, 2], [40, 41, 300]),      # 5 chunks x 4 taps
]


@pytest.mark.parametrize("case", P8X_CASES, ids=lambda c: "%dx%d_ctx%s" % (c[0], c[1], "_".join(map(str, c[2]))))
@pytest.mark.parametrize("mode", ["f32x", "f32xb"])
@pytest.mark.parametrize("act", ["relu", None])
def test_p8x_kernel_vs_oracle_and_x3(case, mode, act, monkeypatch):
    """Round 5: kernels_tdnn_p8x.hip (the persistent 8-phase structure for the f32x mode: f32 rows and [hi | lo] weight rows through
    LDS-DMA, the split in registers, three matrix instructions per product) through the C ABI, forced onto small batches
    (ASV_AMD_P8X=2): against the f64 numpy oracle of TdnnAffine + ReLU + eval BN (components.py:107-149, 418-431) at the mode's
    tolerance AND bit for bit against tdnn_gemm_x3_kernel (the same products into each accumulator in the same order, the same
    epilogue expression) - so an utterance's embedding does not depend on which of the two its batch was sent to."""
    from libs.amd import capi
    L = capi.lib()
    cin, cout, ctx, lens = case
    r = np.random.RandomState(cin * 13 + cout)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    x = r.randn(int(offsets[-1]), cin).astype(np.float32)
    left, right = min(0, ctx[0]), max(0, ctx[-1])
    w = (r.randn(cout, cin, right - left + 1) / np.sqrt(cin * len(ctx))).astype(np.float32)
    b = (0.1 * r.randn(cout)).astype(np.float32)
    scale = r.uniform(0.5, 1.5, cout).astype(np.float32)
    shift = (0.2 * r.randn(cout)).astype(np.float32)
    prec, flags = _mode_args(mode + "_mfma")
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")
    monkeypatch.setenv("ASV_AMD_P8X", "2")
    n0 = L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8X)
    got = _tdnn_forward(x, offsets, w, b, ctx, act, scale, shift, False, prec, flags)
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8X) == n0 + 1, "the f32x 8-phase kernel did not take this layer"
    monkeypatch.setenv("ASV_AMD_P8X", "0")
    ref = _tdnn_forward(x, offsets, w, b, ctx, act, scale, shift, False, prec, flags)
    assert L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8X) == n0 + 1
    assert np.array_equal(got, ref), "f32x 8-phase kernel and tdnn_gemm_x3_kernel differ in %d values (max %g)" % (
        int((got != ref).sum()), float(np.abs(got - ref).max()))
    want = _oracle_layer(x, offsets, w, b, ctx, act, scale, shift, False)
    assert rel_err(got, want) < MODE_TOL[mode]


def test_p8_kernel_is_the_one_the_wide_layers_run_on(monkeypatch):
    """Production dispatch: from one round of 256 x 256 tiles (256 of them) a plain wide layer goes to the 8-phase kernel, smaller
    batches and the layers it does not take (cin = 80: no whole 64-channel chunks) stay on the variant-3 kernel."""
    from libs.amd import capi
    monkeypatch.setenv("ASV_AMD_LIVE_TUNE", "1")     # (the library reads ASV_AMD_P8 once per process otherwise: an earlier test's value would stick)
    monkeypatch.setenv("ASV_AMD_P8", "1")            # 1 = the production rule
    L = capi.lib()
    r = np.random.RandomState(3)
    prec, flags = _mode_args("bf16_mfma")

    def run(cin, cout, ctx, lens):
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        x = r.randn(int(offsets[-1]), cin).astype(np.float32)
        w = (r.randn(cout, cin, max(0, ctx[-1]) - min(0, ctx[0]) + 1) / np.sqrt(cin * len(ctx))).astype(np.float32)
        a, b = L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8), L.asv_kernel_launch_count(capi.KERNEL_TDNN_BIG3)
        _tdnn_forward(x, offsets, w, None, ctx, "relu", None, None, False, prec, flags)
        return L.asv_kernel_launch_count(capi.KERNEL_TDNN_P8) - a, L.asv_kernel_launch_count(capi.KERNEL_TDNN_BIG3) - b

    assert run(512, 512, [-2, 0, 2], [200] * 256) == (1, 0)          # configs[1]'s tdnn2: 408 tiles
    assert run(512, 512, [-2, 0, 2], [200] * 64) == (0, 1)           # 102 tiles: less than a round
    assert run(80, 512, [-2, -1, 0, 1, 2], [200] * 256) == (0, 1)    # cin = 80: no whole chunks
    assert run(192, 512, [0], [200] * 256) == (1, 0)                 # 3 K-tiles: an odd count is fine
