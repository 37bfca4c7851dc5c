# -*- coding:utf-8 -*-
"""Compiles a recorded layer program (ir.Graph) into an `asv_net_t` of libasv_amd.so and runs
batched extraction on device-resident feature matrices.

PyTorch is used here for device memory (tensors as allocations), streams and, in shard.py,
torch.distributed - plumbing only.  All arithmetic happens inside libasv_amd.so.
"""

import ctypes as C
import os

import numpy as np

from . import capi
from . import ir as _ir


PRECISIONS = {"f32": capi.PREC_F32, "fp32": capi.PREC_F32, "float32": capi.PREC_F32,
              "bf16": capi.PREC_BF16, "bfloat16": capi.PREC_BF16,
              "f16": capi.PREC_F16, "fp16": capi.PREC_F16, "half": capi.PREC_F16, "float16": capi.PREC_F16,
              "f32x": capi.PREC_F32X, "bf16x3": capi.PREC_F32X, "f16x3": capi.PREC_F32X, "f32m": capi.PREC_F32X}
H16_MODES = ("bf16", "bfloat16", "f16", "fp16", "half", "float16")       # 16-bit frames-domain storage

# options of the f32x mode, appended with '-' (e.g. "f32x-bf16"): the 16-bit type the operands are split into, and the
# reduced-product MEASUREMENT variants ("why not two matrix instructions per product", LABLOG.md "Precision modes")
X3_OPTIONS = {"bf16": capi.FLAG_X3_SPLIT_BF16, "f16": capi.FLAG_X3_SPLIT_F16, "half": capi.FLAG_X3_SPLIT_F16,
              "noxlo": capi.FLAG_X3_NO_XLO, "nowlo": capi.FLAG_X3_NO_WLO, "mx": capi.FLAG_X3_MX8}


def parse_precision(name):
    """'f32x-bf16-noxlo' -> ('f32x', capi.PREC_F32X, flag bits).  'bf16x3' / 'f16x3' name the split type themselves."""
    parts = name.lower().split("-")
    base = parts[0]
    if base not in PRECISIONS:
        raise ValueError("unknown precision %r (have %s)" % (name, sorted(set(PRECISIONS))))
    # 'f32m' = 'f32x-mx': the f32x mode with its two correction products on the block-scaled 8-bit matrix instruction (round 6)
    bits = {"bf16x3": capi.FLAG_X3_SPLIT_BF16, "f16x3": capi.FLAG_X3_SPLIT_F16, "f32m": capi.FLAG_X3_MX8}.get(base, 0)
    for opt in parts[1:]:
        if PRECISIONS[base] != capi.PREC_F32X or opt not in X3_OPTIONS:
            raise ValueError("precision %r: option %r (only the f32x mode has options: %s)" % (name, opt, sorted(X3_OPTIONS)))
        bits |= X3_OPTIONS[opt]
    return base, PRECISIONS[base], bits


def default_precision():
    """'f32m' (the default since round 6): f32 storage, matrix products on the matrix cores from operands split into IEEE-half hi + lo halves
    (22 significant bits per operand) - the main product w_hi x_hi on the 16-bit instruction, the two corrections w_hi x_lo + w_lo x_hi as ONE
    block-scaled 8-bit instruction per 32 channels (kernels_tdnn_chainm.hip / kernels_tdnn_x3m.hip; kernels without that form run all three
    products on the 16-bit instruction): ~1e-5 of the reference's embeddings, inside the 1e-4 gate and the 0.01 % EER gate on every draw of
    tests/gate_table.py, at 1.35 x the rate of 'f32x' on the x-vector.  'f32x': all three products on the 16-bit instruction (~1e-7; ~3 x the
    rate of 'f32', the exact f32-input MFMA and bit-for-bit fma chain).  'bf16' / 'f16' are the throughput modes (16-bit storage and products,
    f32 accumulate, f32 pooled tail; 'f16' rounds 8x finer at the same rate, operands within +-65504; their measured distance from the
    gates: tests/test_gpu_eer_gate.py)."""
    return os.environ.get("ASV_AMD_PRECISION", "f32m").lower()


def default_flags():
    flags = 0
    if os.environ.get("ASV_AMD_REF_KERNELS", "0") not in ("0", "", "false"):
        flags |= capi.FLAG_REF_KERNELS
    if os.environ.get("ASV_AMD_NO_FUSE", "0") not in ("0", "", "false"):
        flags |= capi.FLAG_NO_FUSE
    if os.environ.get("ASV_AMD_NO_CHAIN", "0") not in ("0", "", "false"):
        flags |= capi.FLAG_NO_CHAIN
    if os.environ.get("ASV_AMD_SMALL_TILES", "0") not in ("0", "", "false"):
        flags |= capi.FLAG_SMALL_TILES
    return flags


def _view_args(v):
    return (-1, 0) if v is None else (v.tid, v.ch_off)


def gather_fuse_on():
    """ASV_AMD_NO_GATHER_FUSE=1 keeps the stage-closing elementwise pass of a ResNet and the strided gathers behind it as two passes
    (A/B switch; the fused form gives the same bits - tests/test_gpu_resnet.py)."""
    return os.environ.get("ASV_AMD_NO_GATHER_FUSE", "0") in ("0", "", "false")


class Engine(object):
    """One compiled model on one device."""

    def __init__(self, graph, device_index=0, precision=None, flags=None):
        self.lib = capi.lib()
        self.graph = graph
        self.device_index = int(device_index)
        self.precision = (precision or default_precision()).lower()
        self.precision_base, prec_id, prec_bits = parse_precision(self.precision)
        self.flags = (default_flags() if flags is None else int(flags)) | prec_bits
        self.embed_dim = graph.output.channels
        self.feat_dim = graph.feat_dim
        self._net = C.c_void_p()
        capi.check(self.lib.asv_net_create(C.byref(self._net), self.device_index, prec_id, self.flags, graph.feat_dim), "asv_net_create")
        try:
            self._build()
        except Exception:
            self.close()
            raise

    # ---- program upload ---------------------------------------------------------------
    def _build(self):
        L, g = self.lib, self.graph
        buf_of = {0: 0}
        dom_of = {0: capi.DOMAIN_FRAMES, 1: capi.DOMAIN_UTTS}
        for i, spec in enumerate(g.domains):
            if spec[0] == "grid":
                dom_of[i] = capi.check(L.asv_net_define_grid(self._net, spec[1], spec[2], spec[3]), "asv_net_define_grid")
            elif spec[0] == "seq":                   # one row per frame at the grid's rate = a grid of width 1, pitch 1
                dom_of[i] = capi.check(L.asv_net_define_grid(self._net, spec[1], 1, 1), "asv_net_define_grid")
        # one device buffer per IR tensor that something writes as a whole or in slices
        # 16-bit engines run every Res2NetBlock as one kernel (kernels_res2.hip); the parity modes keep one layer per branch
        fuse = self.precision_base in H16_MODES and (self.flags & (capi.FLAG_REF_KERNELS | capi.FLAG_NO_FUSE | capi.FLAG_SMALL_TILES)) == 0
        ops = g.fused_res2_ops() if fuse else g.ops
        if (self.flags & (capi.FLAG_REF_KERNELS | capi.FLAG_NO_FUSE)) == 0:
            ops = g.fused_add_ops(ops)               # exact in every precision mode (see its docstring)
            if gather_fuse_on():
                ops = g.fused_gather_ops(ops)        # likewise: the stage-closing elementwise pass as the prologue of the strided gathers
        self.ops = ops                               # the program as uploaded (op indices of the profiling rows refer to it)
        written = sorted({op.out.tid for op in ops} | {op.out2.tid for op in ops if getattr(op, "out2", None) is not None})
        for tid in written:
            dom, ch = g.tensors[tid]
            buf_of[tid] = capi.check(L.asv_net_new_buffer(self._net, dom_of[dom], ch), "asv_net_new_buffer")

        def bv(v):
            if v is None:
                return (-1, 0)
            return (buf_of[v.tid], v.ch_off)

        for op in ops:
            if op.kind == "res2":
                d = capi.Res2Desc()
                d.struct_size = C.sizeof(capi.Res2Desc)
                d.in_buf, d.in_ch_off = bv(op.inp)
                d.out_buf, d.out_ch_off = bv(op.out)
                d.branches, d.dilation = op.branches, op.dilation
                keep = [op.weight, op.bias, op.scale, op.shift]
                d.weight, d.bias, d.scale, d.shift = (capi.f32_ptr(a) for a in keep)
                capi.check(L.asv_net_add_res2(self._net, C.byref(d)), "asv_net_add_res2")
                del keep
            elif op.kind == "tdnn":
                d = capi.TdnnDesc()
                d.struct_size = C.sizeof(capi.TdnnDesc)
                d.in_buf, d.in_ch_off = bv(op.inp)
                d.in2_buf, d.in2_ch_off = bv(op.inp2)
                d.out_buf, d.out_ch_off = bv(op.out)
                d.in_ch, d.out_ch = op.inp.channels, op.out.channels
                d.n_taps = len(op.taps)
                for i, t in enumerate(op.taps):
                    d.taps[i] = t
                keep = [op.weight, op.bias, op.scale, op.shift]
                d.weight = capi.f32_ptr(op.weight)
                d.w_tot_context, d.w_left_context = op.weight.shape[2], op.w_left
                d.bias = capi.f32_ptr(op.bias) if op.bias is not None else None
                d.seg_bias_buf = bv(op.seg_bias)[0]
                d.act1 = capi.ACT_BY_NAME[op.act1]
                d.scale = capi.f32_ptr(op.scale) if op.scale is not None else None
                d.shift = capi.f32_ptr(op.shift) if op.shift is not None else None
                d.affine_first = int(op.affine_first)
                d.act2 = capi.ACT_BY_NAME[op.act2]
                d.seg_scale_buf = bv(op.seg_scale)[0]
                d.res_buf, d.res_ch_off = bv(op.res)
                d.alg_fraction = float(getattr(op, "alg_fraction", 0.0) or 0.0)
                capi.check(L.asv_net_add_tdnn(self._net, C.byref(d)), "asv_net_add_tdnn")
                del keep
            elif op.kind == "pool":
                d = capi.PoolDesc()
                d.struct_size = C.sizeof(capi.PoolDesc)
                d.in_buf, d.in_ch_off = bv(op.inp)
                d.channels = op.inp.channels
                d.out_buf, d.out_ch_off = bv(op.out)
                d.stddev, d.unbiased, d.var_mode, d.eps = int(op.stddev), op.unbiased, op.var_mode, op.eps
                d.per_bin = int(getattr(op, "per_bin", False))
                capi.check(L.asv_net_add_stats_pool(self._net, C.byref(d)), "asv_net_add_stats_pool")
            elif op.kind == "attpool":
                d = capi.AttPoolDesc()
                d.struct_size = C.sizeof(capi.AttPoolDesc)
                d.x_buf, d.x_ch_off = bv(op.x)
                d.logit_buf, d.logit_ch_off = bv(op.logits)
                d.channels = op.x.channels
                d.out_buf, d.out_ch_off = bv(op.out)
                d.eps = op.eps
                d.shared_logits = int(getattr(op, "shared", False))
                d.logit_group = int(getattr(op, "group", 0))
                d.logit_softplus2 = int(getattr(op, "softplus2", False))
                if getattr(op, "prior_logit", None) is not None:
                    d.prior_logit = op.prior_logit.ctypes.data_as(capi.c_float_p)
                    d.prior_value = op.prior_value.ctypes.data_as(capi.c_float_p)
                capi.check(L.asv_net_add_attentive_pool(self._net, C.byref(d)), "asv_net_add_attentive_pool")
            elif op.kind == "lde":
                d = capi.LdeDesc()
                d.struct_size = C.sizeof(capi.LdeDesc)
                d.x_buf, d.x_ch_off = bv(op.x)
                d.channels, d.n_centres = op.x.channels, len(op.beta)
                d.out_buf, d.out_ch_off = bv(op.out)
                d.mu = op.mu.ctypes.data_as(capi.c_float_p)
                d.beta = op.beta.ctypes.data_as(capi.c_float_p)
                capi.check(L.asv_net_add_lde_pool(self._net, C.byref(d)), "asv_net_add_lde_pool")
            elif op.kind == "eltwise":
                d = capi.EltwiseDesc()
                d.struct_size = C.sizeof(capi.EltwiseDesc)
                d.channels = op.a.channels
                d.a_buf, d.a_ch_off = bv(op.a)
                d.b_buf, d.b_ch_off = bv(op.b)
                d.c_buf, d.c_ch_off = bv(op.c)
                d.seg_scale_buf = bv(op.seg_scale)[0]
                d.out_buf, d.out_ch_off = bv(op.out)
                d.scale = capi.f32_ptr(op.scale) if op.scale is not None else None
                d.shift = capi.f32_ptr(op.shift) if op.shift is not None else None
                d.act = capi.ACT_BY_NAME[getattr(op, "act", None)]
                d.seg_norm_buf = bv(getattr(op, "seg_norm", None))[0]
                d.seg_norm_mode = int(getattr(op, "seg_norm_mode", 0))
                d.d_buf, d.d_ch_off = bv(getattr(op, "d", None))
                d.out2_buf, d.out2_ch_off = bv(getattr(op, "out2", None))
                capi.check(L.asv_net_add_eltwise(self._net, C.byref(d)), "asv_net_add_eltwise")
            elif op.kind == "grid_input":
                d = capi.GridInputDesc()
                d.struct_size = C.sizeof(capi.GridInputDesc)
                d.out_buf = buf_of[op.out.tid]
                d.in_buf = buf_of[op.inp.tid]
                capi.check(L.asv_net_add_grid_input(self._net, C.byref(d)), "asv_net_add_grid_input")
            elif op.kind == "flatten":
                d = capi.GridFlattenDesc()
                d.struct_size = C.sizeof(capi.GridFlattenDesc)
                d.in_buf, d.out_buf = buf_of[op.inp.tid], buf_of[op.out.tid]
                capi.check(L.asv_net_add_grid_flatten(self._net, C.byref(d)), "asv_net_add_grid_flatten")
            elif op.kind == "im2col":
                d = capi.Im2colDesc()
                d.struct_size = C.sizeof(capi.Im2colDesc)
                d.in_buf, d.out_buf = buf_of[op.inp.tid], buf_of[op.out.tid]
                d.channels, d.n_taps, d.stride = op.inp.channels, len(op.taps), op.stride
                for i, (dt, df) in enumerate(op.taps):
                    d.dt[i], d.df[i] = dt, df
                d.b_buf = bv(getattr(op, "b", None))[0]
                d.seg_scale_buf = bv(getattr(op, "seg_scale", None))[0]
                d.act = capi.ACT_BY_NAME[getattr(op, "act", None)]
                capi.check(L.asv_net_add_im2col(self._net, C.byref(d)), "asv_net_add_im2col")
            else:
                raise _ir.TraceError("op kind %r survived graph optimisation" % op.kind)
        if g.output.ch_off != 0:
            raise _ir.TraceError("the embedding must start at channel 0 of its buffer")
        capi.check(L.asv_net_finalize(self._net, buf_of[g.output.tid], g.output.channels), "asv_net_finalize")

    def close(self):
        twin = getattr(self, "_wide_range_twin", None)
        if twin is not None:
            self._wide_range_twin = None
            twin.close()
        if getattr(self, "_net", None) is not None and self._net.value:
            self.lib.asv_net_destroy(self._net)
            self._net = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def describe(self):
        buf = C.create_string_buffer(1 << 16)
        capi.check(self.lib.asv_net_describe(self._net, buf, len(buf)), "asv_net_describe")
        return buf.value.decode()

    def device_bytes(self):
        return int(self.lib.asv_net_device_bytes(self._net))

    # ---- extraction ---------------------------------------------------------------------
    def extract_device(self, feats, offsets, max_chunk=10000, out=None, stream=None):
        """feats: torch float32 CUDA tensor [sum T, D] (contiguous, on this engine's device);
        offsets: int32 numpy [B+1].  Returns a torch CUDA tensor [B, E] (asynchronous on the
        current torch stream unless `stream` is given)."""
        import torch
        assert feats.is_cuda and feats.dtype == torch.float32 and feats.is_contiguous(), "feats must be a contiguous float32 CUDA tensor"
        assert feats.device.index == self.device_index, "feats live on cuda:%s, engine on cuda:%d" % (feats.device.index, self.device_index)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        n = len(offsets) - 1
        assert feats.dim() == 2 and feats.shape[1] == self.feat_dim, "expected [frames, %d] features, got %s" % (self.feat_dim, tuple(feats.shape))
        assert int(offsets[-1]) == feats.shape[0], "offsets[-1]=%d but feats has %d rows" % (int(offsets[-1]), feats.shape[0])
        if out is None:
            out = torch.empty((n, self.embed_dim), dtype=torch.float32, device=feats.device)
        if stream is None:
            stream = torch.cuda.current_stream(feats.device).cuda_stream
        capi.check(self.lib.asv_net_extract(self._net, C.c_void_p(feats.data_ptr()), offsets.ctypes.data_as(capi.c_int32_p), n,
                                            C.c_void_p(out.data_ptr()), int(max_chunk), C.c_void_p(stream)), "asv_net_extract")
        return out

    def extract_device_guarded(self, feats, offsets, max_chunk=10000, out=None):
        """extract_device() + the range guard of extract_batch(), for synchronous callers that hold their batch on the device
        (extract_embeddings_online.py): waits for the batch, and re-runs it on the bf16-halves twin with a RuntimeWarning if an
        activation left the range of the f32x mode's operand split."""
        watch = self._range_fallback_applies()
        if watch:
            self.status()                                 # clear what earlier asynchronous calls may have left
        out = self.extract_device(feats, offsets, max_chunk=max_chunk, out=out)
        if watch and out.numel() and (self.status() & capi.STATUS_HALF_RANGE):
            import warnings
            warnings.warn("asv-subtools_amd: an activation left the IEEE-half range of the f32x mode's operand split (|x| > 65504 or NaN): "
                          "re-running the batch with bf16 operand halves (precision 'f32x-bf16')", RuntimeWarning)
            out = self.wide_range_twin().extract_device(feats, offsets, max_chunk=max_chunk, out=out)
        return out

    def extract_batch(self, mats, max_chunk=10000):
        """mats: list of [T_i, D] array-likes (host), or of CUDA tensors (e.g. libs.amd.frontend.fbank output - packed on
        the device, no host round trip).  Returns a CPU float32 tensor [B, E].

        Range guard of the default mode: 'f32x' splits every operand into IEEE-half halves, so an activation beyond +-65504
        (un-normalised features of huge magnitude, a checkpoint whose BatchNorm lets a layer blow up) cannot be represented -
        the f32 reference has no such limit - and the NaN products it causes are mapped to 0 by the next ReLU: wrong embeddings
        without a trace.  The split kernels therefore watch the range of every hi half they produce and raise a status bit
        (asv_net_status); a batch that raised it is re-run ONCE on a lazily compiled twin of this engine with bf16 halves
        ('f32x-bf16': 16 significant bits per operand, the whole f32 exponent range; ~6e-6 relative, still inside the 1e-4
        gate), with a warning.  extract_device() (device tensor out, no host synchronisation) does not check: call status(), or
        status_async() behind it - libs.amd.pipeline.DeviceSets, the path of the extraction scripts, does that for every batch."""
        import torch
        watch = self._range_fallback_applies()
        if watch:
            self.status()                                 # clear what earlier asynchronous calls may have left
        out = self._extract_batch(mats, max_chunk)
        if watch and out.numel() and (self.status() & capi.STATUS_HALF_RANGE):
            import warnings
            warnings.warn("asv-subtools_amd: an activation left the IEEE-half range of the f32x mode's operand split (|x| > 65504 or NaN): "
                          "re-running the batch with bf16 operand halves (precision 'f32x-bf16')", RuntimeWarning)
            out = self.wide_range_twin()._extract_batch(mats, max_chunk)
        return out

    def wide_range_twin(self):
        """The lazily compiled twin of an f32x engine with bf16 operand halves ('f32x-bf16'): what a batch that raised
        STATUS_HALF_RANGE is re-run on (here and in libs.amd.pipeline.DeviceSets, the scripts' path)."""
        if getattr(self, "_wide_range_twin", None) is None:
            self._wide_range_twin = Engine(self.graph, device_index=self.device_index, precision="f32x-bf16",
                                           flags=self.flags & ~(capi.FLAG_X3_SPLIT_F16 | capi.FLAG_X3_SPLIT_BF16 | capi.FLAG_X3_MX8))
        return self._wide_range_twin

    def status(self, stream=None):
        """Range status bits since the last call (capi.STATUS_HALF_RANGE); synchronises on `stream` (default: the current one)."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(torch.device("cuda", self.device_index)).cuda_stream
        word = C.c_uint(0)
        capi.check(self.lib.asv_net_status(self._net, C.byref(word), C.c_void_p(stream)), "asv_net_status")
        return int(word.value)

    def status_async(self, host_word, stream=None):
        """Enqueues the copy of the range status bits into `host_word` (a page-locked int32 / uint32 tensor of one element) behind
        the work already on `stream` (default: the current one) and clears them on the device - no wait: the word is valid once
        an event recorded behind this call has completed.  One batch per engine between two calls."""
        import torch
        assert host_word.is_pinned() and host_word.numel() == 1 and host_word.element_size() == 4
        if stream is None:
            stream = torch.cuda.current_stream(torch.device("cuda", self.device_index)).cuda_stream
        capi.check(self.lib.asv_net_status_async(self._net, C.c_void_p(host_word.data_ptr()), C.c_void_p(stream)), "asv_net_status_async")

    def _range_fallback_applies(self):
        return self.precision_base in ("f32x", "f16x3", "f32m") and (self.flags & (capi.FLAG_X3_SPLIT_BF16 | capi.FLAG_REF_KERNELS)) == 0

    def _extract_batch(self, mats, max_chunk=10000):
        import torch
        mats = list(mats)
        if not mats:                                  # an empty feature archive is not an error in the reference's loop either
            return torch.empty((0, self.embed_dim), dtype=torch.float32)
        if all(isinstance(m, torch.Tensor) and m.is_cuda for m in mats):
            dev = torch.device("cuda", self.device_index)
            for m in mats:
                if m.dim() != 2 or m.shape[1] != self.feat_dim:
                    raise ValueError("expected [frames, %d] feature matrices, got %s" % (self.feat_dim, tuple(m.shape)))
            offsets = np.zeros(len(mats) + 1, dtype=np.int32)
            np.cumsum([m.shape[0] for m in mats], out=offsets[1:])
            with torch.cuda.device(dev):
                packed = torch.cat([m.to(dev, torch.float32) for m in mats], dim=0) if len(mats) > 1 else mats[0].to(dev, torch.float32).contiguous()
                return self.extract_device(packed, offsets, max_chunk=max_chunk).cpu()
        mats = [np.asarray(m.cpu() if isinstance(m, torch.Tensor) else m, dtype=np.float32) for m in mats]
        for m in mats:
            if m.ndim != 2 or m.shape[1] != self.feat_dim:
                raise ValueError("expected [frames, %d] feature matrices, got %s" % (self.feat_dim, m.shape))
        offsets = np.zeros(len(mats) + 1, dtype=np.int32)
        np.cumsum([m.shape[0] for m in mats], out=offsets[1:])
        packed = torch.from_numpy(np.ascontiguousarray(np.concatenate(mats, axis=0)))
        dev = torch.device("cuda", self.device_index)
        with torch.cuda.device(dev):
            out = self.extract_device(packed.to(dev), offsets, max_chunk=max_chunk)
            return out.cpu()

    # ---- profiling ------------------------------------------------------------------------
    def set_profiling(self, enable):
        """0 off, 1 per kernel class, 2 per program op."""
        capi.check(self.lib.asv_net_set_profiling(self._net, int(enable)), "asv_net_set_profiling")

    def get_profile(self):
        """[{name, launches, total_ms, flops}] of the launches since the last call (hipEvents on
        the extract stream); synchronises on the recorded events."""
        cap = 16 + 2 * len(self.graph.ops)
        rows = (capi.KernelTime * cap)()
        n = C.c_int(0)
        capi.check(self.lib.asv_net_get_profile(self._net, rows, cap, C.byref(n)), "asv_net_get_profile")
        return [dict(name=rows[i].name.decode(), op_index=int(rows[i].op_index), launches=int(rows[i].launches), total_ms=float(rows[i].total_ms),
                     flops=float(rows[i].flops)) for i in range(n.value)]


def compile_model(model, function=None, device_index=None, precision=None, flags=None, feat_dim=None):
    """Records `function` (default: the body of the model's decorated extract_embedding) and
    compiles it for `device_index` (default: the device the model's parameters live on)."""
    import torch
    if function is None:
        function = getattr(type(model).extract_embedding, "__wrapped_body__", None)
        if function is None:
            raise _ir.TraceError("%s.extract_embedding is not decorated with libs.nnet.for_extract_embedding" % type(model).__name__)
    if device_index is None:
        p = next(model.parameters())
        if not p.is_cuda:
            raise RuntimeError("the model lives on %s: asv-subtools_amd extracts on a ROCm device only (select_model_device(model, "
                               "use_gpu='true')); there is no CPU fallback" % p.device)
        device_index = p.device.index if p.device.index is not None else torch.cuda.current_device()
    if feat_dim is None:
        feat_dim = infer_feat_dim(model)
    graph = _ir.trace(model, function, feat_dim)
    return Engine(graph, device_index=device_index, precision=precision, flags=flags)


def infer_feat_dim(model):
    """Input feature dimension = in-channels of the first TDNN / conv layer."""
    d = getattr(model, "inputs_dim", None)
    if d is not None:
        return int(d)
    for m in model.modules():
        if hasattr(m, "input_dim") and hasattr(m, "context"):
            return int(m.input_dim)
    if d is None:
        raise _ir.TraceError("cannot infer the feature dimension of %s; pass feat_dim" % type(model).__name__)
    return int(d)
