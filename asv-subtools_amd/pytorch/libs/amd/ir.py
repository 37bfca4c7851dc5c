# -*- coding:utf-8 -*-
"""Layer-program IR and the symbolic recorder that turns a model blueprint's own
`extract_embedding` body into it.

The reference executes a blueprint eagerly, one torch op at a time, one utterance at a time
(libs/nnet/framework.py:18-52).  Here the same Python body is run ONCE on a `Sym` handle:
the `libs.nnet` layer classes (and a few handlers for torch-owned modules) append coarse,
already-fused ops to a `Graph`, a few graph passes remove copies (concat elision, add
folding), and `engine.py` hands the result to libasv_amd.so through the C ABI.  Anything
the recorder does not understand raises - there is no eager fallback.

Tensor semantics mirror the reference's [batch, channels, frames] convention:
  frames-domain Sym: rank 3, shape (1, C, T)      utts-domain Sym: rank 3 (1, C, 1) or rank 2 (1, C)
"""

import numpy as np

DOMAIN_FRAMES, DOMAIN_UTTS = 0, 1
CHAN_ALIGN = 16
MAX_HALO = 4


class TraceError(NotImplementedError):
    """The blueprint used something the MI355X path does not implement (raised, never bypassed)."""


class View(object):
    """Channel slice [ch_off, ch_off+channels) of IR tensor `tid`."""
    __slots__ = ("tid", "ch_off", "channels")

    def __init__(self, tid, ch_off, channels):
        self.tid, self.ch_off, self.channels = int(tid), int(ch_off), int(channels)

    def key(self):
        return (self.tid, self.ch_off, self.channels)

    def __eq__(self, other):
        return isinstance(other, View) and self.key() == other.key()

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self.key())

    def __repr__(self):
        return "t%d[%d:+%d]" % (self.tid, self.ch_off, self.channels)


class Op(object):
    """kind in {'tdnn','pool','attpool','eltwise','cat','grid_input','im2col'}; `out` is a View covering
    a whole tensor until concat elision redirects it into a slice of a wider one."""

    def __init__(self, kind, out, **attrs):
        self.kind, self.out = kind, out
        self.__dict__.update(attrs)

    def inputs(self):
        if self.kind == "tdnn":
            names = ("inp", "inp2", "seg_bias", "seg_scale", "res")
        elif self.kind == "pool":
            names = ("inp",)
        elif self.kind == "attpool":
            names = ("x", "logits")
        elif self.kind == "lde":
            names = ("x",)
        elif self.kind == "eltwise":
            names = ("a", "b", "c", "seg_scale", "seg_norm", "d")
        elif self.kind == "im2col":
            names = ("inp", "b", "seg_scale")
        elif self.kind in ("grid_input", "res2", "flatten"):
            names = ("inp",)
        else:
            return list(self.parts)
        return [getattr(self, n) for n in names if getattr(self, n, None) is not None]

    def input_names(self):
        return {"tdnn": ("inp", "inp2", "seg_bias", "seg_scale", "res"), "pool": ("inp",),
                "attpool": ("x", "logits"), "lde": ("x",), "eltwise": ("a", "b", "c", "seg_scale", "seg_norm", "d"), "cat": (), "grid_input": ("inp",),
                "im2col": ("inp", "b", "seg_scale"), "res2": ("inp",), "flatten": ("inp",)}[self.kind]


class Graph(object):
    def __init__(self, feat_dim):
        self.tensors = []          # [(domain, channels)]
        self.ops = []
        self.feat_dim = int(feat_dim)
        # row domains: 0 frames, 1 utts, >= 2 (time, frequency) grids ("grid", time_shift, width, pitch) and the sequences
        # ("seq", time_shift) their flattened [B, C*F, T'] views live on (one row per frame of the subsampled time axis)
        self.domains = [("frames",), ("utts",)]
        self.new_tensor(DOMAIN_FRAMES, feat_dim)        # tensor 0 = input features
        self.output = None                                # View (utts domain)

    def new_tensor(self, domain, channels):
        self.tensors.append((int(domain), int(channels)))
        return len(self.tensors) - 1

    def domain(self, tid):
        return self.tensors[tid][0]

    def full_view(self, tid):
        return View(tid, 0, self.tensors[tid][1])

    def grid_domain(self, shift, width, pitch=None):
        spec = ("grid", int(shift), int(width), int(pitch if pitch is not None else width + 1))
        if spec[3] + 1 > 84:
            raise TraceError("2-D trunk: %d frequency bins exceed the widest window the MI355X conv kernel stages (82)" % width)
        if spec not in self.domains:
            self.domains.append(spec)
        return self.domains.index(spec)

    def seq_domain(self, shift):
        spec = ("seq", int(shift))
        if spec not in self.domains:
            self.domains.append(spec)
        return self.domains.index(spec)

    def seq_spec(self, tid):
        d = self.domains[self.domain(tid)]
        return d if d[0] == "seq" else None

    def is_utts(self, tid):
        return self.domains[self.domain(tid)][0] == "utts"

    def grid_spec(self, tid):
        d = self.domains[self.domain(tid)]
        return d if d[0] == "grid" else None

    # ---- op constructors -------------------------------------------------------------
    def tdnn(self, inp, weight, bias, taps, w_left, act1=None, scale=None, shift=None, affine_first=False,
             act2=None, inp2=None, seg_bias=None, seg_scale=None, res=None):
        weight = np.ascontiguousarray(weight, dtype=np.float32)
        assert weight.ndim == 3 and weight.shape[1] == inp.channels, (weight.shape, inp)
        taps = [int(t) for t in taps]
        dom = self.domain(inp.tid)
        halo = MAX_HALO if dom == DOMAIN_FRAMES else (self.domains[dom][3] + 1 if self.domains[dom][0] == "grid" else (2 if self.domains[dom][0] == "seq" else 0))
        if max(abs(t) for t in taps) > halo and dom != DOMAIN_UTTS:
            raise TraceError("TDNN context %s reaches beyond the +-%d row halo of the MI355X row layout" % (taps, halo))
        if dom == DOMAIN_UTTS and taps != [0]:
            raise TraceError("a pooled (utterance-level) tensor only supports context [0], got %s" % (taps,))
        out = self.full_view(self.new_tensor(dom, weight.shape[0]))
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
        self.ops.append(Op("tdnn", out, inp=inp, inp2=inp2, weight=weight, bias=f32(bias), taps=taps, w_left=int(w_left),
                           act1=act1 or None, scale=f32(scale), shift=f32(shift), affine_first=bool(affine_first),
                           act2=act2 or None, seg_bias=seg_bias, seg_scale=seg_scale, res=res))
        return out

    def pool(self, inp, stddev=True, unbiased=0, var_mode=0, eps=1e-10, per_bin=False):
        ch = inp.channels * (2 if stddev else 1)
        if per_bin:
            ch *= self.grid_spec(inp.tid)[2]
        out = self.full_view(self.new_tensor(DOMAIN_UTTS, ch))
        self.ops.append(Op("pool", out, inp=inp, stddev=bool(stddev), unbiased=int(unbiased), var_mode=int(var_mode), eps=float(eps),
                           per_bin=bool(per_bin)))
        return out

    def grid_input(self, inp=None):
        """features [T][F] (raw input or a normalised copy) -> one-channel (time, frequency) grid."""
        inp = inp or self.full_view(0)
        dom = self.grid_domain(0, self.feat_dim)
        out = self.full_view(self.new_tensor(dom, 1))
        self.ops.append(Op("grid_input", out, inp=inp))
        return out

    def im2col(self, inp, taps, stride):
        """taps: [(dt, df)]; output grid = input grid subsampled by `stride` in both axes."""
        g = self.grid_spec(inp.tid)
        assert g is not None and inp.ch_off == 0 and inp.channels == self.tensors[inp.tid][1]
        if inp.channels % CHAN_ALIGN != 0:
            raise TraceError("strided 2-D convolution over %d channels: the gather needs a multiple of %d" % (inp.channels, CHAN_ALIGN))
        dom = self.grid_domain(g[1] + (1 if stride == 2 else 0), (g[2] + stride - 1) // stride)
        out = self.full_view(self.new_tensor(dom, inp.channels * len(taps)))
        self.ops.append(Op("im2col", out, inp=inp, taps=[(int(a), int(b)) for a, b in taps], stride=int(stride)))
        return out

    def flatten_grid(self, inp):
        """[B, C, F', T'] -> [B, C*F', T'] with the reference's channel index c*F' + f (resnet_xvector.py:193), as a tensor on the
        sequence domain of the grid's time shift: what the frame-weighting poolings and frame-level layers behind the 2-D trunk read."""
        g = self.grid_spec(inp.tid)
        assert g is not None and inp.ch_off == 0 and inp.channels == self.tensors[inp.tid][1]
        out = self.full_view(self.new_tensor(self.seq_domain(g[1]), inp.channels * g[2]))
        self.ops.append(Op("flatten", out, inp=inp))
        return out

    def attpool(self, x, logits, eps=1e-5, shared=False, group=0, softplus2=False, prior_logit=None, prior_value=None):
        """softmax-over-frames weighted mean / std of x -> [mean | std].  `shared`: logits has ONE channel that weights every
        channel of x; `group` > 1: every `group` consecutive channels of x share logit column (channel // group).
        xi-vector options (per-channel logits): `softplus2` turns the stored values z into logits 2 log(softplus(z));
        `prior_logit` / `prior_value` [channels] add one more frame with these logits (untransformed) and values."""
        group = int(group)
        if (softplus2 or prior_logit is not None) and (shared or group > 1):
            raise TraceError("the xi-vector pooling options need per-channel logits")
        n_logits = 1 if shared else (-(-x.channels // group) if group > 1 else x.channels)
        if logits.channels != n_logits:
            raise TraceError("attention logits must have %d channel(s), got %d" % (n_logits, logits.channels))
        out = self.full_view(self.new_tensor(DOMAIN_UTTS, 2 * x.channels))
        f32 = lambda v: None if v is None else np.ascontiguousarray(v, dtype=np.float32).reshape(-1)
        self.ops.append(Op("attpool", out, x=x, logits=logits, eps=float(eps), shared=bool(shared), group=group, softplus2=bool(softplus2),
                           prior_logit=f32(prior_logit), prior_value=f32(prior_value)))
        return out

    def lde(self, x, mu, beta):
        """Learnable dictionary encoding pooling: mu [channels, centres], beta [centres] -> [channels * centres] per utterance
        (column c * centres + k)."""
        mu = np.ascontiguousarray(mu, dtype=np.float32)
        beta = np.ascontiguousarray(beta, dtype=np.float32).reshape(-1)
        if mu.shape != (x.channels, beta.shape[0]) or not 1 <= beta.shape[0] <= 64:
            raise TraceError("lde: mu %s / beta %s do not fit %d channels (at most 64 centres)" % (mu.shape, beta.shape, x.channels))
        out = self.full_view(self.new_tensor(DOMAIN_UTTS, x.channels * beta.shape[0]))
        self.ops.append(Op("lde", out, x=x, mu=mu, beta=beta))
        return out

    def eltwise(self, a, b=None, c=None, seg_scale=None, scale=None, shift=None, act=None, seg_norm=None, seg_norm_mode=0):
        out = self.full_view(self.new_tensor(self.domain(a.tid), a.channels))
        f32 = lambda v: None if v is None else np.ascontiguousarray(v, dtype=np.float32)
        self.ops.append(Op("eltwise", out, a=a, b=b, c=c, seg_scale=seg_scale, scale=f32(scale), shift=f32(shift), act=act or None,
                           seg_norm=seg_norm, seg_norm_mode=int(seg_norm_mode)))
        return out

    def cat(self, parts, align=1):
        """Channel concatenation; with `align` every part starts at a multiple of it (the gaps are never written and stay
        zero in the device arena: the consumer gives them zero weights, see Sym.col_order)."""
        dom = self.domain(parts[0].tid)
        width = sum(-(-p.channels // align) * align for p in parts[:-1]) + parts[-1].channels
        out = self.full_view(self.new_tensor(dom, width))
        self.ops.append(Op("cat", out, parts=list(parts), align=int(align)))
        return out

    # ---- passes ----------------------------------------------------------------------
    def _use_count(self):
        uses = {}
        for op in self.ops:
            for v in op.inputs():
                uses[v.tid] = uses.get(v.tid, 0) + 1
        if self.output is not None:
            uses[self.output.tid] = uses.get(self.output.tid, 0) + 1
        return uses

    def _producer(self):
        return {op.out.tid: op for op in self.ops if op.out.ch_off == 0 and op.out.channels == self.tensors[op.out.tid][1]}

    def _replace_tensor(self, old_tid, new_view):
        """Every read of tensor `old_tid` (whole or sliced) now reads inside `new_view`."""
        def fix(v):
            if v is not None and v.tid == old_tid:
                return View(new_view.tid, new_view.ch_off + v.ch_off, v.channels)
            return v
        for op in self.ops:
            if op.kind == "cat":
                op.parts = [fix(v) for v in op.parts]
            else:
                for n in op.input_names():
                    setattr(op, n, fix(getattr(op, n, None)))
        self.output = fix(self.output)

    def optimize(self):
        self._cse_adds()
        self._fold_eltwise_chains()
        self._fold_adds_into_tdnn()
        self._elide_cats()
        self._drop_dead()
        return self

    def _is_plain_add(self, op):
        return (op.kind == "eltwise" and op.b is not None and op.c is None and op.seg_scale is None and op.scale is None
                and getattr(op, "act", None) is None and getattr(op, "seg_norm", None) is None)

    def _cse_adds(self):
        seen = {}
        for op in list(self.ops):
            if not self._is_plain_add(op):
                continue
            key = tuple(sorted([op.a.key(), op.b.key()]))
            if key in seen:
                self._replace_tensor(op.out.tid, seen[key].out)
                self.ops.remove(op)
            else:
                seen[key] = op

    def _fold_eltwise_chains(self):
        """(a*s + b) + c  ->  one eltwise, when the intermediate has a single reader."""
        changed = True
        while changed:
            changed = False
            uses, prod = self._use_count(), self._producer()
            for op in self.ops:
                if not self._is_plain_add(op):
                    continue
                for first, other in ((op.a, op.b), (op.b, op.a)):
                    p = prod.get(first.tid)
                    if (p is None or p.kind != "eltwise" or p is op or uses.get(first.tid, 0) != 1 or getattr(p, "act", None) is not None
                            or first.channels != self.tensors[first.tid][1] or first.ch_off != 0):
                        continue
                    slot = "b" if p.b is None else ("c" if p.c is None else None)
                    if slot is None:
                        continue
                    setattr(p, slot, other)
                    self._replace_tensor(op.out.tid, p.out)
                    self.ops.remove(op)
                    # keep program order valid: p must run after `other`'s producer
                    self._move_after_inputs(p)
                    changed = True
                    break
                if changed:
                    break

    def _move_after_inputs(self, op):
        prod = self._producer()
        idx = self.ops.index(op)
        need = max([self.ops.index(prod[v.tid]) for v in op.inputs() if v.tid in prod] + [-1])
        if need > idx:
            self.ops.remove(op)
            self.ops.insert(need, op)        # after removal, index `need` is right behind the last input producer

    def _fold_adds_into_tdnn(self):
        """tdnn(a + b) -> tdnn(inp=a, inp2=b): the add happens while the window is staged."""
        uses, prod = self._use_count(), self._producer()
        for op in list(self.ops):
            if op.kind != "tdnn" or op.inp2 is not None:
                continue
            p = prod.get(op.inp.tid)
            if p is None or not self._is_plain_add(p) or uses.get(op.inp.tid, 0) != 1:
                continue
            if op.inp.ch_off != 0 or op.inp.channels != self.tensors[op.inp.tid][1]:
                continue
            op.inp, op.inp2 = p.a, p.b
            self.ops.remove(p)

    def _elide_cats(self):
        """torch.cat over channels becomes "producers write into slices of the wide buffer";
        parts that are views of something else (Res2Net's pass-through group) get one copy."""
        for op in list(self.ops):
            if op.kind != "cat":
                continue
            prod = self._producer()
            idx = self.ops.index(op)
            off = 0
            new_ops = []
            for part in op.parts:
                whole = part.ch_off == 0 and part.channels == self.tensors[part.tid][1]
                p = prod.get(part.tid)
                dst = View(op.out.tid, off, part.channels)
                if whole and p is not None and p.kind != "cat" and part.tid != 0 and off % CHAN_ALIGN == 0:
                    p.out = dst
                    self._replace_tensor(part.tid, dst)
                else:
                    new_ops.append(Op("eltwise", dst, a=part, b=None, c=None, seg_scale=None, scale=None, shift=None, act=None,
                                      seg_norm=None, seg_norm_mode=0))
                align = getattr(op, "align", 1)
                off += -(-part.channels // align) * align
            self.ops[idx:idx + 1] = new_ops

    def _drop_dead(self):
        changed = True
        while changed:
            changed = False
            live = set()
            for op in self.ops:
                for v in op.inputs():
                    live.add(v.tid)
            if self.output is not None:
                live.add(self.output.tid)
            for op in list(self.ops):
                if op.out.tid not in live:
                    self.ops.remove(op)
                    changed = True

    def fused_res2_ops(self, width=128, max_branches=7):
        """The op list with every Res2NetBlock (ecapa_tdnn_xvector.py:61-75 - after cat elision: n dependent `width`-channel
        TDNN ops `y_k = f(y_{k-1} + x_k)` writing slices of one buffer plus the pass-through copy of group 0) replaced by ONE
        'res2' op (kernels_res2.hip).  Engines use it in bf16 mode; anything that does not match exactly is left alone."""
        ops, out, i = self.ops, [], 0
        plain = lambda o: (o.kind == "tdnn" and o.act1 == "relu" and o.scale is not None and not o.affine_first and o.act2 is None
                           and o.seg_bias is None and o.seg_scale is None and o.res is None and o.bias is not None)
        while i < len(ops):
            first = ops[i]
            grp = None
            if (plain(first) and first.inp2 is None and first.inp.channels == width and first.out.channels == width and len(first.taps) == 3
                    and first.taps[1] == 0 and first.taps[0] == -first.taps[2] and 1 <= first.taps[2] <= MAX_HALO and first.inp.ch_off == width
                    and first.out.ch_off == width and self.domain(first.inp.tid) == DOMAIN_FRAMES and first.inp.tid != first.out.tid):
                H, O, d = first.inp.tid, first.out.tid, first.taps[2]
                n = self.tensors[H][1] // width - 1
                ok = (self.tensors[H][1] == (n + 1) * width and self.tensors[O][1] == (n + 1) * width and 1 <= n <= max_branches and i + n < len(ops)
                      and first.weight.shape == (width, width, 2 * d + 1) and first.w_left == -d)
                k = 2
                while ok and k <= n:
                    o = ops[i + k - 1]
                    ok = (plain(o) and o.taps == first.taps and o.inp == View(O, (k - 1) * width, width) and o.inp2 == View(H, k * width, width)
                          and o.out == View(O, k * width, width) and o.weight.shape == first.weight.shape and o.w_left == -d)
                    k += 1
                if ok:
                    cp = ops[i + n]
                    ok = (cp.kind == "eltwise" and cp.a == View(H, 0, width) and cp.out == View(O, 0, width) and cp.b is None and cp.c is None
                          and cp.seg_scale is None and cp.scale is None and getattr(cp, "act", None) is None and getattr(cp, "seg_norm", None) is None)
                if ok:
                    br = ops[i:i + n]
                    grp = Op("res2", View(O, 0, (n + 1) * width), inp=View(H, 0, (n + 1) * width), branches=n, dilation=d,
                             weight=np.ascontiguousarray(np.stack([o.weight for o in br]), dtype=np.float32),
                             bias=np.ascontiguousarray(np.stack([o.bias for o in br]), dtype=np.float32),
                             scale=np.ascontiguousarray(np.stack([o.scale for o in br]), dtype=np.float32),
                             shift=np.ascontiguousarray(np.stack([o.shift for o in br]), dtype=np.float32))
            if grp is not None:
                out.append(grp)
                i += grp.branches + 1
            else:
                out.append(first)
                i += 1
        return out

    def fused_add_ops(self, ops=None):
        """The op list with every plain addition `q = p.out + w` that directly follows its producer `p` (an eltwise op)
        folded into `p` as a second output (`out2`, addend `d`): ECAPA's running sum of block outputs
        (ecapa_tdnn_xvector.py:262-268: out_k = SE_Res2Block(...); next input = previous input + out_k) then costs two
        reads and two writes per block instead of three + three.  The second output adds to the value as STORED (rounded
        to the buffer's element type), so every bit of both tensors is what the two separate passes produce."""
        ops = list(self.ops if ops is None else ops)
        out = []
        position = {}
        for idx, op in enumerate(ops):
            position.setdefault(op.out.tid, idx)
        skip = set()
        for idx, op in enumerate(ops):
            if idx in skip:
                continue
            nxt = ops[idx + 1] if idx + 1 < len(ops) else None
            if (op.kind == "eltwise" and getattr(op, "out2", None) is None and nxt is not None and self._is_plain_add(nxt)
                    and self.domain(op.out.tid) == self.domain(nxt.out.tid)):
                other = nxt.b if nxt.a == op.out else (nxt.a if nxt.b == op.out else None)
                ready = other is not None and other.tid != op.out.tid and (other.tid == 0 or position.get(other.tid, len(ops)) < idx)
                if ready and other.channels == op.out.channels and nxt.out.tid != op.out.tid:
                    fused = Op("eltwise", op.out, **{k: v for k, v in op.__dict__.items() if k not in ("kind", "out")})
                    fused.d, fused.out2 = other, nxt.out
                    out.append(fused)
                    skip.add(idx + 1)
                    continue
            out.append(op)
        return out

    def fused_gather_ops(self, ops=None):
        """The op list with every elementwise pass `o = act(a * seg_scale + b)` that only im2col gathers read folded into those
        gathers as their prologue: the output of the last block of a ResNet stage (resnet.py:70-85: relu(se(y) + identity)) feeds
        nothing but the stride-2 convolutions of the next stage (their gather, and the 1x1 downsample's), so the tensor is never
        needed in its own layout.  One pass less over the widest maps of the stage; the prologue performs the elementwise kernel's
        operations in its order with its roundings (kernels_pool.hip), so every bit of the gathered tensor stays what the two
        passes produce."""
        ops = list(self.ops if ops is None else ops)
        readers = {}
        for idx, op in enumerate(ops):
            for v in op.inputs():
                readers.setdefault(v.tid, []).append(idx)
        fold = {}                                            # index of a gather -> the eltwise op folded into it
        drop = set()
        for idx, op in enumerate(ops):
            if op.kind != "eltwise" or op.a is None or op.act not in (None, "relu"):
                continue
            if any(getattr(op, n, None) is not None for n in ("c", "scale", "shift", "seg_norm", "d", "out2")):
                continue
            if getattr(op, "b", None) is None and getattr(op, "seg_scale", None) is None and op.act is None:
                continue
            whole = lambda v: v is None or (v.ch_off == 0 and v.channels == self.tensors[v.tid][1])
            if not (whole(op.out) and whole(op.a) and whole(getattr(op, "b", None))):
                continue
            if self.output is not None and self.output.tid == op.out.tid:
                continue
            rd = readers.get(op.out.tid, [])
            if not rd or any(ops[r].kind != "im2col" or ops[r].inp.tid != op.out.tid or getattr(ops[r], "b", None) is not None
                             or getattr(ops[r], "seg_scale", None) is not None or getattr(ops[r], "act", None) is not None for r in rd):
                continue
            if any(o.out.tid == op.out.tid for j, o in enumerate(ops) if j != idx):      # written in slices by somebody else
                continue
            # the operands must still hold their values where the gathers run: nothing between rewrites them (buffers are
            # written once per pass in this IR, so a later writer of the same tensor id is the only hazard)
            last = max(rd)
            srcs = {v.tid for v in op.inputs()}
            if any(o.out.tid in srcs for o in ops[idx + 1:last + 1]):
                continue
            drop.add(idx)
            for r in rd:
                fold[r] = op
        out = []
        for idx, op in enumerate(ops):
            if idx in drop:
                continue
            if idx in fold:
                e = fold[idx]
                g = Op("im2col", op.out, **{k: v for k, v in op.__dict__.items() if k not in ("kind", "out")})
                g.inp, g.b, g.seg_scale, g.act = e.a, getattr(e, "b", None), getattr(e, "seg_scale", None), e.act
                out.append(g)
            else:
                out.append(op)
        return out

    def describe(self):
        lines = ["graph: %d tensors, %d ops, output %r" % (len(self.tensors), len(self.ops), self.output)]
        for i, op in enumerate(self.ops):
            if op.kind == "tdnn":
                extra = "taps=%s act1=%s affine=%s%s act2=%s" % (op.taps, op.act1, op.scale is not None, "(first)" if op.affine_first else "", op.act2)
                ins = "%r%s" % (op.inp, ("+%r" % op.inp2) if op.inp2 is not None else "")
                for n in ("seg_bias", "seg_scale", "res"):
                    if getattr(op, n) is not None:
                        extra += " %s=%r" % (n, getattr(op, n))
            elif op.kind == "pool":
                ins, extra = repr(op.inp), "stddev=%s unbiased=%d var_mode=%d eps=%g per_bin=%s" % (op.stddev, op.unbiased, op.var_mode, op.eps, op.per_bin)
            elif op.kind == "grid_input":
                ins, extra = repr(op.inp), ""
            elif op.kind == "im2col":
                ins, extra = repr(op.inp), "taps=%d stride=%d" % (len(op.taps), op.stride)
            elif op.kind == "flatten":
                ins, extra = repr(op.inp), ""
            elif op.kind == "attpool":
                ins, extra = "x=%r logits=%r" % (op.x, op.logits), "eps=%g" % op.eps
            elif op.kind == "lde":
                ins, extra = "x=%r" % (op.x,), "centres=%d" % len(op.beta)
            elif op.kind == "eltwise":
                ins = " ".join("%s=%r" % (n, getattr(op, n)) for n in ("a", "b", "c", "seg_scale") if getattr(op, n) is not None)
                extra = "affine=%s act=%s" % (op.scale is not None, op.act)
            else:
                ins, extra = repr(op.parts), ""
            lines.append("  %2d %-8s %s -> %r  %s" % (i, op.kind, ins, op.out, extra))
        return "\n".join(lines)

    def flops_per_frame(self):
        """Algorithmic 2*MAC per input frame of the frames-domain layers + per-utterance rest
        (active taps only)."""
        per_frame = per_utt = 0
        for op in self.ops:
            if op.kind != "tdnn":
                continue
            f = 2 * op.inp.channels * op.weight.shape[0] * len(op.taps) * (getattr(op, "alg_fraction", None) or 1.0)
            g = self.grid_spec(op.inp.tid)
            if self.domain(op.inp.tid) == DOMAIN_FRAMES:
                per_frame += f
            elif g is not None:
                per_frame += f * g[2] / float(1 << g[1])       # width positions per frame, every 2^shift-th frame
            elif self.seq_spec(op.inp.tid) is not None:
                per_frame += f / float(1 << self.seq_spec(op.inp.tid)[1])
            else:
                per_utt += f
        return per_frame, per_utt


# ------------------------------------------------------------------------------------------
# symbolic tensor handed to the blueprint's extract_embedding body

class Sym(object):
    """rank 3: [1, C, T] (frames) / [1, C, 1] (pooled); rank 2: [1, C] (pooled); rank 4: [1, C, F, T] on a
    (time, frequency) grid.  `flat_grid` marks the [1, C*F, T] reshape of a rank-4 tensor; `col_order`
    (pooled tensors) maps this tensor's columns to the reference's column order."""

    def __init__(self, graph, view, rank=3, flat_grid=False, col_order=None):
        self.graph, self.view, self.rank = graph, view, rank
        self.flat_grid, self.col_order = flat_grid, col_order

    def as_sequence(self):
        """self, or - for the [1, C*F, T] reshape of a grid tensor - the materialised [T'][c*F + f] tensor (once per Sym)."""
        if not self.flat_grid:
            return self
        if getattr(self, "_flat", None) is None:
            self._flat = Sym(self.graph, self.graph.flatten_grid(self.view), 3)
        return self._flat

    # -- what blueprint code inspects
    @property
    def domain(self):
        return self.graph.domain(self.view.tid)

    @property
    def shape(self):
        g = self.graph.grid_spec(self.view.tid)
        if g is not None:
            return (1, self.view.channels * g[2], -1) if self.flat_grid else (1, self.view.channels, g[2], -1)
        if self.rank == 2:
            return (1, self.view.channels)
        return (1, self.view.channels, 1 if self.domain == DOMAIN_UTTS else -1)

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return self.rank

    @property
    def device(self):
        return "hip-symbolic"

    def __len__(self):
        return 1

    # -- shape ops that are free in this layout
    def unsqueeze(self, dim):
        if self.rank == 2 and dim in (2, -1) and self.domain == DOMAIN_UTTS:
            return Sym(self.graph, self.view, 3, col_order=self.col_order)
        if (self.rank == 3 and dim == 1 and self.domain == DOMAIN_FRAMES and self.view.ch_off == 0
                and self.view.channels == self.graph.feat_dim == self.graph.tensors[self.view.tid][1]):
            # ResNetXvector: [B, F, T] -> [B, 1, F, T] (resnet_xvector.py:191)
            return Sym(self.graph, self.graph.grid_input(self.view), 4)
        raise TraceError("unsqueeze(%r) of a rank-%d %s tensor is not supported" % (dim, self.rank, "utts" if self.domain else "frames"))

    def squeeze(self, dim=None):
        if self.rank == 3 and self.domain == DOMAIN_UTTS and dim in (None, 2, -1):
            return Sym(self.graph, self.view, 2, col_order=self.col_order)
        raise TraceError("squeeze(%r) is only supported on pooled tensors" % (dim,))

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        g = self.graph.grid_spec(self.view.tid)
        if g is not None and self.rank == 4 and len(shape) == 3 and shape[1] == self.view.channels * g[2]:
            return Sym(self.graph, self.view, 3, flat_grid=True)       # [B, C*F, T]: channel index c*F + f
        raise TraceError("reshape%r of this tensor is not supported on the MI355X path" % (tuple(shape),))

    def contiguous(self):
        return self

    def float(self):
        return self

    def to(self, *a, **k):
        return self

    # -- arithmetic
    def __add__(self, other):
        if not isinstance(other, Sym):
            if isinstance(other, (int, float)) and other == 0:
                return self
            raise TraceError("only tensor + tensor is supported on the HIP path (got %r)" % type(other))
        if other.view.channels != self.view.channels or other.domain != self.domain:
            raise TraceError("add of mismatched tensors %r + %r" % (self.view, other.view))
        return Sym(self.graph, self.graph.eltwise(self.view, b=other.view), max(self.rank, other.rank))

    __radd__ = __add__

    def chunk(self, chunks, dim=1):
        return sym_chunk(self, chunks, dim)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        handler = _torch_handlers().get(func)
        if handler is None:
            raise TraceError("torch function %s is not implemented by the MI355X extract path "
                             "(supported: cat, chunk, add, conv1d(kernel=1), batch_norm(eval), unsqueeze, squeeze)"
                             % getattr(func, "__name__", func))
        return handler(*args, **(kwargs or {}))


def sym_chunk(x, chunks, dim=1):
    if dim != 1:
        raise TraceError("chunk along dim %r: only the channel dim (1) is supported" % (dim,))
    C = x.view.channels
    size = (C + chunks - 1) // chunks
    out, off = [], 0
    while off < C:
        n = min(size, C - off)
        if (x.view.ch_off + off) % CHAN_ALIGN != 0:
            raise TraceError("channel chunk at offset %d is not %d-aligned" % (off, CHAN_ALIGN))
        out.append(Sym(x.graph, View(x.view.tid, x.view.ch_off + off, n), x.rank))
        off += n
    return tuple(out)


def sym_cat(tensors, dim=0, **kw):
    tensors = list(tensors)
    if dim != 1 or not all(isinstance(t, Sym) for t in tensors):
        raise TraceError("cat: only channel-dim (1) concatenation of traced tensors is supported")
    g = tensors[0].graph
    return Sym(g, g.cat([t.view for t in tensors]), tensors[0].rank)


def fold_batchnorm(running_mean, running_var, weight, bias, eps):
    """eval BatchNorm -> per-channel (scale, shift), folded in float64."""
    rm = np.asarray(running_mean, dtype=np.float64)
    rv = np.asarray(running_var, dtype=np.float64)
    scale = 1.0 / np.sqrt(rv + float(eps))
    if weight is not None:
        scale = scale * np.asarray(weight, dtype=np.float64)
    shift = -rm * scale
    if bias is not None:
        shift = shift + np.asarray(bias, dtype=np.float64)
    return scale.astype(np.float32), shift.astype(np.float32)


def _np(t):
    return None if t is None else t.detach().cpu().numpy()


def sym_batch_norm(x, running_mean, running_var, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
    if training or running_mean is None:
        raise TraceError("batch_norm in training mode / without running stats cannot be traced")
    scale, shift = fold_batchnorm(_np(running_mean), _np(running_var), _np(weight), _np(bias), eps)
    order = getattr(x, "col_order", None)
    if order is not None:          # a pooled tensor whose columns are a permutation (with gaps) of the reference's: permute the constants with it
        if int((order >= 0).sum()) != scale.shape[0]:
            raise TraceError("batch_norm over %d channels applied to a tensor of %d" % (scale.shape[0], int((order >= 0).sum())))
        sc, sh = np.ones(order.shape[0], dtype=scale.dtype), np.zeros(order.shape[0], dtype=shift.dtype)
        sc[order >= 0], sh[order >= 0] = scale[order[order >= 0]], shift[order[order >= 0]]
        return Sym(x.graph, x.graph.eltwise(x.view, scale=sc, shift=sh), x.rank, col_order=order)
    return Sym(x.graph, x.graph.eltwise(x.view, scale=scale, shift=shift), x.rank)


def sym_conv1d(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    one = lambda v: v[0] if isinstance(v, (tuple, list)) else v
    if weight.shape[2] != 1 or one(stride) != 1 or one(padding) != 0 or groups != 1:
        raise TraceError("raw conv1d: only kernel_size=1, stride=1, padding=0, groups=1 is supported; "
                         "use libs.nnet.TdnnAffine for context")
    return Sym(x.graph, x.graph.tdnn(x.view, _np(weight), _np(bias), [0], 0), 3)


_HANDLERS = None


def _torch_handlers():
    global _HANDLERS
    if _HANDLERS is None:
        import torch
        import torch.nn.functional as F
        _HANDLERS = {
            torch.cat: sym_cat, torch.concat: sym_cat, torch.concatenate: sym_cat,
            torch.chunk: sym_chunk, torch.Tensor.chunk: sym_chunk,
            torch.add: lambda a, b, **k: a + b, torch.Tensor.add: lambda a, b, **k: a + b,
            F.batch_norm: sym_batch_norm, torch.batch_norm: sym_batch_norm,
            F.conv1d: sym_conv1d, torch.conv1d: sym_conv1d,
            torch.unsqueeze: lambda x, dim: x.unsqueeze(dim), torch.squeeze: lambda x, dim=None: x.squeeze(dim),
        }
    return _HANDLERS


# ------------------------------------------------------------------------------------------
# handlers for modules that model blueprints define themselves (recognised by class name,
# semantics per /root/reference/pytorch/model/ecapa_tdnn_xvector.py)

def _handle_se_connect(mod, x):
    """SE_Connect (ecapa_tdnn_xvector.py:97-111): AdaptiveAvgPool1d(1) -> Conv1d -> ReLU ->
    Conv1d -> Sigmoid -> channel scale."""
    import torch
    seq = mod.se
    convs = [m for m in seq if isinstance(m, torch.nn.Conv1d)]
    if len(convs) != 2 or any(c.kernel_size != (1,) for c in convs):
        raise TraceError("SE_Connect with an unexpected layout: %r" % (seq,))
    g = x.graph
    m = g.pool(x.view, stddev=False)
    h = g.tdnn(m, _np(convs[0].weight), _np(convs[0].bias), [0], 0, act1="relu")
    s = g.tdnn(h, _np(convs[1].weight), _np(convs[1].bias), [0], 0, act1="sigmoid")
    return Sym(g, g.eltwise(x.view, seg_scale=s), 3)


def _handle_attentive_stats_pool(mod, x):
    """AttentiveStatsPool (ecapa_tdnn_xvector.py:156-188).  The 3C->bottleneck conv over
    [x ; mean ; std] is split: the x part runs per frame, the (mean, std) part is constant
    over an utterance and is hoisted into a per-utterance bias."""
    import torch
    att = mod.attention
    conv1, bn, conv2 = att[0], att[2], att[4]
    if not (isinstance(conv1, torch.nn.Conv1d) and isinstance(bn, torch.nn.BatchNorm1d) and isinstance(conv2, torch.nn.Conv1d)
            and isinstance(att[1], torch.nn.ReLU) and isinstance(att[3], torch.nn.Tanh) and isinstance(att[5], torch.nn.Softmax)):
        raise TraceError("AttentiveStatsPool with an unexpected attention stack: %r" % (att,))
    g = x.graph
    C = x.view.channels
    w1, b1 = _np(conv1.weight), _np(conv1.bias)
    scale, shift = fold_batchnorm(_np(bn.running_mean), _np(bn.running_var), _np(bn.weight) if bn.affine else None,
                                  _np(bn.bias) if bn.affine else None, bn.eps)
    if mod.time_attention:
        assert w1.shape[1] == 3 * C
        gstats = g.pool(x.view, stddev=True, unbiased=2, var_mode=1, eps=1e-5)     # torch.var (unbiased) + 1e-5, 176-178
        ctx = g.tdnn(gstats, w1[:, C:, :], b1, [0], 0)                              # W_mean.mean + W_std.std + b
        h = g.tdnn(x.view, w1[:, :C, :], None, [0], 0, seg_bias=ctx, act1="relu", scale=scale, shift=shift, act2="tanh")
    else:
        h = g.tdnn(x.view, w1, b1, [0], 0, act1="relu", scale=scale, shift=shift, act2="tanh")
    e = g.tdnn(h, _np(conv2.weight), _np(conv2.bias), [0], 0)
    return Sym(g, g.attpool(x.view, e, eps=1e-5), 2)


MODULE_HANDLERS = {
    "SE_Connect": _handle_se_connect,
    "AttentiveStatsPool": _handle_attentive_stats_pool,
}


def trace(model, function, feat_dim):
    """Runs `function(model, sym_input)` (the blueprint's undecorated extract_embedding) and
    returns the optimised Graph."""
    import types
    g = Graph(feat_dim)
    patched = []
    for mod in model.modules():
        h = MODULE_HANDLERS.get(type(mod).__name__)
        if h is not None and not getattr(type(mod), "_asv_amd_native", False):
            patched.append((mod, mod.__dict__.get("forward", None)))
            mod.forward = types.MethodType(lambda self, x, _h=h: _h(self, x), mod)
    was_training = model.training
    model.eval()                                   # framework.py:24 - extraction always runs in eval mode
    try:
        out = function(model, Sym(g, g.full_view(0), 3))
    finally:
        if was_training:
            model.train()
        for mod, old in patched:
            if old is None:
                del mod.forward
            else:
                mod.forward = old
    if not isinstance(out, Sym) or out.domain != DOMAIN_UTTS:
        raise TraceError("extract_embedding must return a pooled (utterance-level) tensor produced by libs.nnet layers")
    g.output = out.view
    return g.optimize()
