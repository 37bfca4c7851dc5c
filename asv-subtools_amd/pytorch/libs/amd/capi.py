# -*- coding:utf-8 -*-
"""ctypes binding of libasv_amd.so (include/asv_amd.h).

The shared library is the product: there is NO Python/torch compute fallback.  If the
library is missing or a call fails, an exception is raised - loudly - and the caller's
extraction job exits non-zero exactly like the reference script does on any error
(pipeline/onestep/extract_embeddings.py:85-88).
"""

import ctypes as C
import os

ASV_OK = 0
PREC_F32, PREC_BF16, PREC_F32X, PREC_F16 = 0, 1, 2, 3
FLAG_REF_KERNELS, FLAG_NO_FUSE, FLAG_SMALL_TILES, FLAG_BIG_V2, FLAG_NO_CHAIN = 1, 2, 4, 8, 16
FLAG_X3_SPLIT_BF16, FLAG_X3_SPLIT_F16, FLAG_X3_NO_XLO, FLAG_X3_NO_WLO, FLAG_X3_TILE128, FLAG_X3_MX8 = 32, 64, 128, 256, 512, 1024
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
DOMAIN_FRAMES, DOMAIN_UTTS = 0, 1
MAX_TAPS = 9
POOL_VAR_CLAMP, POOL_VAR_ADD = 0, 1
PLDA_NORM_NONE, PLDA_NORM_SIMPLE, PLDA_NORM_PSI = 0, 1, 2
STATUS_HALF_RANGE = 1
KERNEL_TDNN_P8, KERNEL_TDNN_BIG3, KERNEL_TDNN_P8X, KERNEL_TDNN_CHAINM, KERNEL_TDNN_X3M, KERNEL_TDNN_X3M_IMAGE = 1, 2, 3, 4, 5, 6

ACT_BY_NAME = {None: ACT_NONE, "": ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "tanh": ACT_TANH,
               "sigmoid": ACT_SIGMOID}

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class TdnnDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("in_buf", C.c_int32), ("in_ch_off", C.c_int32),
        ("in2_buf", C.c_int32), ("in2_ch_off", C.c_int32),
        ("out_buf", C.c_int32), ("out_ch_off", C.c_int32),
        ("in_ch", C.c_int32), ("out_ch", C.c_int32),
        ("n_taps", C.c_int32), ("taps", C.c_int32 * MAX_TAPS),
        ("weight", c_float_p),
        ("w_tot_context", C.c_int32), ("w_left_context", C.c_int32),
        ("bias", c_float_p),
        ("seg_bias_buf", C.c_int32), ("act1", C.c_int32),
        ("scale", c_float_p), ("shift", c_float_p),
        ("affine_first", C.c_int32), ("act2", C.c_int32),
        ("seg_scale_buf", C.c_int32),
        ("res_buf", C.c_int32), ("res_ch_off", C.c_int32),
        ("alg_fraction", C.c_float),
    ]


class PoolDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("in_buf", C.c_int32), ("in_ch_off", C.c_int32), ("channels", C.c_int32),
        ("out_buf", C.c_int32), ("out_ch_off", C.c_int32),
        ("stddev", C.c_int32), ("unbiased", C.c_int32), ("var_mode", C.c_int32),
        ("eps", C.c_float),
        ("per_bin", C.c_int32),
    ]


class AttPoolDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("x_buf", C.c_int32), ("x_ch_off", C.c_int32),
        ("logit_buf", C.c_int32), ("logit_ch_off", C.c_int32), ("channels", C.c_int32),
        ("out_buf", C.c_int32), ("out_ch_off", C.c_int32),
        ("eps", C.c_float),
        ("shared_logits", C.c_int32),
        ("logit_group", C.c_int32),
        ("logit_softplus2", C.c_int32), ("prior_logit", c_float_p), ("prior_value", c_float_p),
    ]


class LdeDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("x_buf", C.c_int32), ("x_ch_off", C.c_int32), ("channels", C.c_int32), ("n_centres", C.c_int32),
        ("out_buf", C.c_int32), ("out_ch_off", C.c_int32),
        ("mu", c_float_p), ("beta", c_float_p),
    ]


class Res2Desc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("in_buf", C.c_int32), ("in_ch_off", C.c_int32),
        ("out_buf", C.c_int32), ("out_ch_off", C.c_int32),
        ("branches", C.c_int32), ("dilation", C.c_int32),
        ("weight", c_float_p), ("bias", c_float_p), ("scale", c_float_p), ("shift", c_float_p),
    ]


class EltwiseDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("channels", C.c_int32),
        ("a_buf", C.c_int32), ("a_ch_off", C.c_int32),
        ("b_buf", C.c_int32), ("b_ch_off", C.c_int32),
        ("c_buf", C.c_int32), ("c_ch_off", C.c_int32),
        ("seg_scale_buf", C.c_int32),
        ("out_buf", C.c_int32), ("out_ch_off", C.c_int32),
        ("scale", c_float_p), ("shift", c_float_p),
        ("act", C.c_int32),
        ("seg_norm_buf", C.c_int32), ("seg_norm_mode", C.c_int32),
        ("d_buf", C.c_int32), ("d_ch_off", C.c_int32), ("out2_buf", C.c_int32), ("out2_ch_off", C.c_int32),
    ]


class GridInputDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("out_buf", C.c_int32), ("in_buf", C.c_int32)]


class GridFlattenDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("in_buf", C.c_int32), ("out_buf", C.c_int32)]


class Im2colDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("in_buf", C.c_int32), ("out_buf", C.c_int32), ("channels", C.c_int32), ("n_taps", C.c_int32), ("stride", C.c_int32),
        ("dt", C.c_int32 * MAX_TAPS), ("df", C.c_int32 * MAX_TAPS),
        ("b_buf", C.c_int32), ("seg_scale_buf", C.c_int32), ("act", C.c_int32),
    ]


class FbankOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("sample_rate", C.c_float), ("frame_length_ms", C.c_float), ("frame_shift_ms", C.c_float), ("preemph", C.c_float),
        ("remove_dc_offset", C.c_int32), ("window_type", C.c_int32), ("round_to_power_of_two", C.c_int32), ("snip_edges", C.c_int32),
        ("num_bins", C.c_int32), ("low_freq", C.c_float), ("high_freq", C.c_float),
        ("use_energy", C.c_int32), ("energy_floor", C.c_float), ("raw_energy", C.c_int32), ("htk_compat", C.c_int32),
        ("use_log_fbank", C.c_int32), ("use_power", C.c_int32),
        ("num_ceps", C.c_int32), ("cepstral_lifter", C.c_float),
        ("blackman_coeff", C.c_float), ("vtln_warp", C.c_float), ("vtln_low", C.c_float), ("vtln_high", C.c_float),
    ]


WINDOW_TYPES = {"povey": 0, "hamming": 1, "hanning": 2, "rectangular": 3, "sine": 4, "blackman": 5}


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("op_index", C.c_int32), ("launches", C.c_int32), ("total_ms", C.c_float), ("flops", C.c_double)]


class AsvError(RuntimeError):
    pass


_LIB = None

# every symbol include/asv_amd.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "asv_version", "asv_last_error", "asv_device_count",
    "asv_net_create", "asv_net_destroy", "asv_net_define_grid", "asv_net_new_buffer", "asv_net_add_tdnn",
    "asv_net_add_grid_input", "asv_net_add_im2col", "asv_net_add_grid_flatten",
    "asv_net_add_stats_pool", "asv_net_add_attentive_pool", "asv_net_add_lde_pool", "asv_net_add_eltwise", "asv_net_add_res2",
    "asv_net_finalize", "asv_net_embed_dim", "asv_net_describe", "asv_net_extract",
    "asv_net_device_bytes", "asv_net_set_profiling", "asv_net_get_profile", "asv_net_status", "asv_net_status_async", "asv_kernel_launch_count",
    "asv_tdnn_forward", "asv_stats_pool_forward",
    "asv_length_norm", "asv_mean_vec", "asv_dot_score_matrix", "asv_dot_score_trials",
    "asv_plda_transform", "asv_plda_llr_trials", "asv_eer", "asv_score_norm", "asv_group_mean", "asv_two_cov_trials",
    "asv_plda_train", "asv_scatter_f64", "asv_class_scatter_f64",
    "asv_fbank_num_frames", "asv_fbank", "asv_fbank_pcm16", "asv_cmvn", "asv_cmvn_sliding", "asv_vad_energy", "asv_select_frames",
]


def library_path():
    env = os.environ.get("ASV_AMD_LIB")
    if env:
        return env
    here = os.path.dirname(os.path.abspath(__file__))
    # .../asv-subtools_amd/pytorch/libs/amd -> .../asv-subtools_amd/libasv_amd.so
    return os.path.normpath(os.path.join(here, "..", "..", "..", "libasv_amd.so"))


def lib():
    """Loads libasv_amd.so once.  Raises AsvError (never falls back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise AsvError("libasv_amd.so not found at %s - build it with `python __graft_entry__.py` "
                       "(or `make -C asv-subtools_amd/csrc`); there is no CPU/torch fallback." % path)
    try:
        L = C.CDLL(path, mode=getattr(C, "RTLD_GLOBAL", 0))
    except OSError as e:
        raise AsvError("could not load %s: %s" % (path, e))
    vp, ci, cu = C.c_void_p, C.c_int, C.c_uint
    L.asv_version.restype = ci
    L.asv_last_error.restype = C.c_char_p
    L.asv_device_count.argtypes = [C.POINTER(ci)]
    L.asv_net_create.argtypes = [C.POINTER(vp), ci, ci, cu, ci]
    L.asv_net_destroy.argtypes = [vp]; L.asv_net_destroy.restype = None
    L.asv_net_new_buffer.argtypes = [vp, ci, ci]
    L.asv_net_define_grid.argtypes = [vp, ci, ci, ci]
    L.asv_net_add_grid_input.argtypes = [vp, C.POINTER(GridInputDesc)]
    L.asv_net_add_im2col.argtypes = [vp, C.POINTER(Im2colDesc)]
    L.asv_net_add_grid_flatten.argtypes = [vp, C.POINTER(GridFlattenDesc)]
    L.asv_net_add_tdnn.argtypes = [vp, C.POINTER(TdnnDesc)]
    L.asv_net_add_stats_pool.argtypes = [vp, C.POINTER(PoolDesc)]
    L.asv_net_add_attentive_pool.argtypes = [vp, C.POINTER(AttPoolDesc)]
    L.asv_net_add_lde_pool.argtypes = [vp, C.POINTER(LdeDesc)]
    L.asv_net_add_eltwise.argtypes = [vp, C.POINTER(EltwiseDesc)]
    L.asv_net_add_res2.argtypes = [vp, C.POINTER(Res2Desc)]
    L.asv_net_finalize.argtypes = [vp, ci, ci]
    L.asv_net_embed_dim.argtypes = [vp]
    L.asv_net_describe.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.asv_net_extract.argtypes = [vp, vp, c_int32_p, ci, vp, ci, vp]
    L.asv_net_device_bytes.argtypes = [vp]; L.asv_net_device_bytes.restype = C.c_size_t
    L.asv_net_status.argtypes = [vp, C.POINTER(C.c_uint), vp]
    L.asv_net_status_async.argtypes = [vp, vp, vp]
    L.asv_kernel_launch_count.argtypes = [ci]; L.asv_kernel_launch_count.restype = C.c_ulonglong
    L.asv_net_set_profiling.argtypes = [vp, ci]
    L.asv_net_get_profile.argtypes = [vp, C.POINTER(KernelTime), ci, C.POINTER(ci)]
    L.asv_tdnn_forward.argtypes = [C.POINTER(TdnnDesc), ci, cu, vp, c_int32_p, ci, vp, vp]
    L.asv_stats_pool_forward.argtypes = [vp, ci, c_int32_p, ci, ci, ci, ci, C.c_float, vp, vp]
    L.asv_length_norm.argtypes = [vp, ci, ci, vp, ci, vp]
    L.asv_mean_vec.argtypes = [vp, ci, ci, vp, vp]
    L.asv_dot_score_matrix.argtypes = [vp, ci, vp, ci, ci, vp, vp]
    L.asv_dot_score_trials.argtypes = [vp, vp, ci, vp, vp, ci, vp, vp]
    L.asv_plda_transform.argtypes = [vp, ci, ci, vp, vp, vp, vp, ci, vp, vp]
    L.asv_plda_llr_trials.argtypes = [vp, vp, ci, vp, vp, vp, vp, ci, vp, vp]
    L.asv_eer.argtypes = [vp, vp, ci, c_float_p, c_float_p, vp]
    L.asv_group_mean.argtypes = [vp, ci, ci, vp, vp, ci, vp, vp, vp]
    L.asv_two_cov_trials.argtypes = [vp, ci, vp, ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), vp, vp, ci, vp, vp]
    L.asv_score_norm.argtypes = [vp, ci, vp, ci, ci, vp, vp, vp, ci, ci, ci, vp, vp]
    L.asv_plda_train.argtypes = [vp, ci, ci, ci, C.POINTER(C.c_int32), C.POINTER(C.c_longlong), ci, ci,
                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    L.asv_scatter_f64.argtypes = [vp, ci, ci, ci, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    L.asv_class_scatter_f64.argtypes = [vp, ci, ci, ci, C.POINTER(C.c_int32), C.POINTER(C.c_longlong), ci,
                                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    L.asv_fbank_num_frames.argtypes = [C.POINTER(FbankOpts), C.c_longlong]; L.asv_fbank_num_frames.restype = C.c_longlong
    L.asv_fbank.argtypes = [C.POINTER(FbankOpts), vp, C.POINTER(C.c_longlong), ci, vp, vp]
    L.asv_fbank_pcm16.argtypes = [C.POINTER(FbankOpts), vp, C.POINTER(C.c_longlong), ci, vp, vp]
    L.asv_cmvn_sliding.argtypes = [vp, vp, C.POINTER(C.c_longlong), ci, ci, ci, ci, ci, ci, vp]
    L.asv_vad_energy.argtypes = [vp, C.POINTER(C.c_longlong), ci, ci, C.c_float, C.c_float, ci, C.c_float, vp, C.POINTER(C.c_longlong), vp]
    L.asv_select_frames.argtypes = [vp, vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), ci, ci, vp, vp]
    L.asv_cmvn.argtypes = [vp, C.POINTER(C.c_longlong), ci, ci, ci, ci, C.c_float, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)          # AttributeError here = header/.so mismatch
        if name not in ("asv_last_error", "asv_net_destroy", "asv_net_device_bytes", "asv_fbank_num_frames"):
            fn.restype = ci
    _LIB = L
    return L


def check(rc, what=""):
    """Turns a negative return code into an exception carrying asv_last_error()."""
    if rc < 0:
        msg = lib().asv_last_error()
        raise AsvError("%s failed (%d): %s" % (what or "libasv_amd call", rc, msg.decode("utf-8", "replace") if msg else ""))
    return rc


def f32_ptr(arr):
    """Host pointer of a C-contiguous float32 numpy array (caller keeps `arr` alive)."""
    import numpy as np
    assert isinstance(arr, np.ndarray) and arr.dtype == np.float32 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(c_float_p)
