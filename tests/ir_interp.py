"""numpy interpreter of a libs.amd.ir.Graph (TEST INFRASTRUCTURE).

Executes the recorded, optimised layer program op by op with the oracle's layer functions,
one utterance at a time.  It checks on CPU what the GPU cannot be reached for: that the
recorder + graph passes (concat elision, add folding, hoisted attention bias) preserve the
reference semantics, before the same program is handed to libasv_amd.so.
"""

import numpy as np

from oracle import np_oracle as O


def _act(x, name):
    return O._act(x, name)


def run_graph(graph, feats, dtype=np.float32, ops=None):
    """feats [T, D] -> embedding [E] for ONE utterance segment (no chunking).  `ops` replaces graph.ops (the engine-level
    late passes fused_res2_ops / fused_add_ops return such lists)."""
    feats = np.asarray(feats, dtype=dtype)
    T = feats.shape[0]
    bufs = {0: feats}

    def grid_dims(tid):
        g = graph.grid_spec(tid)
        fr = T
        for _ in range(g[1]):
            fr = (fr + 1) // 2
        return fr, g[2], g[3]                     # frames, width, pitch

    def seq_frames(tid):
        fr = T
        for _ in range(graph.seq_spec(tid)[1]):
            fr = (fr + 1) // 2
        return fr

    def rows(tid):
        if graph.grid_spec(tid) is not None:
            fr, _, pitch = grid_dims(tid)
            return fr * pitch
        if graph.seq_spec(tid) is not None:
            return seq_frames(tid)
        return T if graph.domain(tid) == 0 else 1

    def valid_mask(tid):
        """[rows, 1] 1.0 where the device's row_valid bit is set (grid: first `width` rows of every frame)."""
        if graph.grid_spec(tid) is None:
            return 1.0
        fr, width, pitch = grid_dims(tid)
        return ((np.arange(fr * pitch) % pitch) < width).astype(dtype)[:, None]

    def get(v):
        return bufs[v.tid][:, v.ch_off:v.ch_off + v.channels]

    def put(v, val):
        if v.tid not in bufs:
            bufs[v.tid] = np.zeros((rows(v.tid), graph.tensors[v.tid][1]), dtype=dtype)
        bufs[v.tid][:, v.ch_off:v.ch_off + v.channels] = val

    for op in (graph.ops if ops is None else ops):
        if op.kind == "tdnn":
            x = get(op.inp)
            if op.inp2 is not None:
                x = x + get(op.inp2)
            z = _tdnn_general(x, op, dtype)
            if op.seg_bias is not None:
                z = z + get(op.seg_bias)
            s = 1 if op.scale is None else op.scale.astype(dtype)
            t = 0 if op.shift is None else op.shift.astype(dtype)
            z = _act(z * s + t, op.act1) if op.affine_first else _act(z, op.act1) * s + t
            z = _act(z, op.act2)
            if op.seg_scale is not None:
                z = z * get(op.seg_scale)
            if op.res is not None:
                z = z + get(op.res)
            put(op.out, (z * valid_mask(op.out.tid)).astype(dtype))
        elif op.kind == "grid_input":
            fr, width, pitch = grid_dims(op.out.tid)
            g = np.zeros((fr * pitch, 1), dtype=dtype)
            src = get(op.inp)
            for t in range(fr):
                g[t * pitch:t * pitch + width, 0] = src[t, :width]
            put(op.out, g)
        elif op.kind == "flatten":
            x = get(op.inp)
            fr, width, pitch = grid_dims(op.inp.tid)
            C = op.inp.channels
            out = np.zeros((fr, C * width), dtype=dtype)
            for f in range(width):
                out[:, f::width] = x[f::pitch][:fr]               # column c*F + f
            put(op.out, out)
        elif op.kind == "im2col":
            x = get(op.inp)
            if getattr(op, "seg_scale", None) is not None:        # the elementwise prologue of Graph.fused_gather_ops
                x = x * get(op.seg_scale)
            if getattr(op, "b", None) is not None:
                x = x + get(op.b)
            x = _act(x, getattr(op, "act", None)).astype(dtype)
            fi, wi, pi = grid_dims(op.inp.tid)
            fo, wo, po = grid_dims(op.out.tid)
            C = op.inp.channels
            out = np.zeros((fo * po, C * len(op.taps)), dtype=dtype)
            for t in range(fo):
                for f in range(wo):
                    for k, (dt, df) in enumerate(op.taps):
                        ti, fi_ = t * op.stride + dt, f * op.stride + df
                        if 0 <= ti < fi and 0 <= fi_ < wi:
                            out[t * po + f, k * C:(k + 1) * C] = x[ti * pi + fi_]
            put(op.out, out)
        elif op.kind == "pool" and getattr(op, "per_bin", False):
            x = get(op.inp)
            fr, width, pitch = grid_dims(op.inp.tid)
            parts = []
            for f in range(width):
                col = x[f::pitch][:fr]
                mean = col.mean(axis=0, dtype=dtype)
                var = ((col - mean) ** 2).sum(axis=0, dtype=dtype) / dtype(fr - 1 if (op.unbiased == 1 and fr > 1) else fr)
                parts += [mean, np.sqrt(np.maximum(var, dtype(op.eps)))] if op.stddev else [mean]
            put(op.out, np.concatenate(parts)[None, :])
        elif op.kind == "pool":
            x = get(op.inp)
            mean = x.mean(axis=0, dtype=dtype)
            if op.stddev:
                n = x.shape[0]
                counts = n - 1 if (op.unbiased == 2 or (op.unbiased == 1 and n > 1)) else n
                with np.errstate(divide="ignore", invalid="ignore"):
                    var = ((x - mean) ** 2).sum(axis=0, dtype=dtype) / dtype(counts)
                std = np.sqrt(var + dtype(op.eps)) if op.var_mode == 1 else np.sqrt(np.maximum(var, dtype(op.eps)))
                put(op.out, np.concatenate([mean, std])[None, :])
            else:
                put(op.out, mean[None, :])
        elif op.kind == "attpool":
            x, e = get(op.x), get(op.logits)                       # e: [T, C], or [T, 1] broadcast over the channels when shared
            if getattr(op, "group", 0) > 1:
                e = e[:, np.arange(x.shape[1]) // op.group]        # heads over channel groups
            if getattr(op, "softplus2", False):                    # xi-vector: precision estimate -> 2 log softplus
                e = dtype(2.0) * np.log(np.where(e > 20, e, np.log1p(np.exp(np.minimum(e, 20)))))
            if getattr(op, "prior_logit", None) is not None:       # the prior is one more frame
                e = np.concatenate([e, op.prior_logit.astype(dtype)[None, :]], axis=0)
                x = np.concatenate([x, op.prior_value.astype(dtype)[None, :]], axis=0)
            a = np.exp(e - e.max(axis=0, keepdims=True))
            a = a / a.sum(axis=0, keepdims=True, dtype=dtype)
            mean = (a * x).sum(axis=0, dtype=dtype)
            resid = (a * x * x).sum(axis=0, dtype=dtype) - mean * mean
            put(op.out, np.concatenate([mean, np.sqrt(np.maximum(resid, dtype(op.eps)))])[None, :])
        elif op.kind == "lde":
            x = get(op.x)                                          # [T, C]
            r = x[:, :, None] - op.mu.astype(dtype)[None, :, :]    # [T, C, K]
            l = -op.beta.astype(dtype)[None, :] * (r * r).sum(axis=1, dtype=dtype)
            w = np.exp(l - l.max(axis=1, keepdims=True))
            w = w / w.sum(axis=1, keepdims=True, dtype=dtype)
            put(op.out, (w[:, None, :] * r).mean(axis=0, dtype=dtype).reshape(1, -1))
        elif op.kind == "eltwise":
            z = get(op.a)
            if op.scale is not None:
                z = z * op.scale.astype(dtype) + op.shift.astype(dtype)
            if getattr(op, "seg_norm", None) is not None:
                st = get(op.seg_norm)[0]
                C = op.a.channels
                if op.seg_norm_mode & 1:
                    z = z - st[:C]
                if op.seg_norm_mode & 2:
                    z = z / st[C:2 * C]
            if op.seg_scale is not None:
                z = z * get(op.seg_scale)
            if op.b is not None:
                z = z + get(op.b)
            if op.c is not None:
                z = z + get(op.c)
            z = _act(z, getattr(op, "act", None))
            put(op.out, (z * valid_mask(op.out.tid)).astype(dtype))
            if getattr(op, "out2", None) is not None:              # second output: the STORED value + d
                put(op.out2, ((get(op.out) + get(op.d)) * valid_mask(op.out2.tid)).astype(dtype))
        elif op.kind == "res2":
            # one Res2NetBlock: group 0 passes through, y_k = affine(relu(conv_k(y_{k-1} + x_k))) with y_0 = 0
            W = op.weight.shape[1]
            x = get(op.inp)
            T = x.shape[0]
            y = np.zeros((T, (op.branches + 1) * W), dtype=dtype)
            y[:, :W] = x[:, :W]
            prev = np.zeros((T, W), dtype=dtype)
            for k in range(1, op.branches + 1):
                u = x[:, k * W:(k + 1) * W] + prev
                z = np.zeros((T, W), dtype=dtype)
                for d in (-op.dilation, 0, op.dilation):           # dense (2d+1)-tap kernels with three live taps
                    lo, hi = max(0, -d), min(T, T - d)
                    if hi > lo:
                        z[lo:hi] += u[lo + d:hi + d] @ op.weight[k - 1][:, :, d + op.dilation].T.astype(dtype)
                z = np.maximum(z + op.bias[k - 1].astype(dtype), 0) * op.scale[k - 1].astype(dtype) + op.shift[k - 1].astype(dtype)
                y[:, k * W:(k + 1) * W] = prev = z.astype(dtype)
            put(op.out, y)
        else:
            raise AssertionError("unexpected op %s" % op.kind)
    return get(graph.output)[0]


def _tdnn_general(x, op, dtype):
    """tap-by-tap evaluation for sliced weights (hoisted attention context etc.)."""
    T = x.shape[0]
    y = np.zeros((T, op.weight.shape[0]), dtype=dtype)
    for d in op.taps:
        k = d - op.w_left
        lo, hi = max(0, -d), min(T, T - d)
        if hi > lo:
            y[lo:hi] += x[lo + d:hi + d] @ op.weight[:, :, k].T.astype(dtype)
    if op.bias is not None:
        y += op.bias.astype(dtype)
    return y


def extract(graph, feats, max_chunk=10000, dtype=np.float32, ops=None):
    return O.extract_embedding(lambda c: run_graph(graph, c, dtype, ops), feats, max_chunk, dtype)
