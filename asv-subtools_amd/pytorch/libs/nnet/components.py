# -*- coding:utf-8 -*-
"""TDNN building blocks with the reference's names, constructor arguments and state_dict
keys (/root/reference/pytorch/libs/nnet/components.py: TdnnAffine 20-165,
_BaseActivationBatchNorm 337-431, ReluBatchNormTdnnLayer 434-461), so reference checkpoints
(`*.params`) load unchanged.

These modules are *parameter holders + program emitters*: called with the symbolic handle of
`libs.amd.ir` they append one fused op (affine + activation + folded eval-BatchNorm) to the
layer program that libasv_amd.so executes.  Calling them eagerly on a torch tensor raises:
training / eager forward is outside this package, and a silent torch fallback would hide a
missing HIP library.
"""

import torch

import libs.support.utils as utils
from libs.amd import ir as _ir
from .activation import Nonlinearity, activation_name


def _eager_unsupported(name):
    raise NotImplementedError(
        "%s.forward() on a torch tensor: eager/training forward is not part of asv-subtools_amd; "
        "call model.extract_embedding(feats) (HIP path) instead" % name)


class TdnnAffine(torch.nn.Module):
    """y = splice(x, context) . W + b; checkpoint weight is the dense [out, in, right-left+1]
    kernel, of which only the taps in `context` are used (reference components.py:78-83)."""

    def __init__(self, input_dim, output_dim, context=[0], bias=True, pad=True, stride=1, groups=1,
                 norm_w=False, norm_f=False):
        super(TdnnAffine, self).__init__()
        assert input_dim % groups == 0
        for a, b in zip(context[:-1], context[1:]):
            if a >= b:
                raise ValueError("Context tuple {} is invalid, such as the order.".format(context))
        self.input_dim, self.output_dim = input_dim, output_dim
        self.context = list(context)
        self.bool_bias, self.pad, self.groups, self.stride = bias, pad, groups, stride
        self.norm_w, self.norm_f = norm_w, norm_f
        self.left_context = context[0] if context[0] < 0 else 0
        self.right_context = context[-1] if context[-1] > 0 else 0
        self.tot_context = self.right_context - self.left_context + 1
        if self.tot_context > 1 and self.norm_f:
            self.norm_f = False
        self.weight = torch.nn.Parameter(torch.empty(output_dim, input_dim // groups, self.tot_context))
        self.bias = torch.nn.Parameter(torch.empty(output_dim)) if bias else None
        # kept for state/attribute compatibility; never multiplied at run time - the packed
        # device weights simply contain the active taps only
        if len(context) != self.tot_context:
            self.mask = torch.tensor([[[1 if i in context else 0 for i in range(self.left_context, self.right_context + 1)]]])
        else:
            self.mask = None
        self.init_weight()

    def init_weight(self):
        torch.nn.init.normal_(self.weight, 0.0, 0.01)
        if self.bias is not None:
            torch.nn.init.constant_(self.bias, 0.0)

    def _check_supported(self):
        if not self.pad or self.stride != 1 or self.norm_f:
            raise _ir.TraceError("TdnnAffine(pad=%s, stride=%s, norm_f=%s): only pad=True, stride=1 without feature "
                                 "normalisation is implemented on the MI355X path (norm_w is: it is folded into the weights)"
                                 % (self.pad, self.stride, self.norm_f))

    def emit(self, x, act1=None, scale=None, shift=None, affine_first=False, row_scale=None, row_map=None):
        """Appends this affine (+ fused epilogue) to the program of the symbolic tensor `x`.  A grouped affine (the attention
        heads of pooling.py:279-296) becomes the block-diagonal dense matrix it is; `row_scale` multiplies output rows
        (weights and bias) by constants."""
        self._check_supported()
        x = x.as_sequence()                    # behind the 2-D trunk: the [B, C*F', T'] reshape, materialised once
        order = getattr(x, "col_order", None)
        have = x.view.channels if order is None else int((order >= 0).sum())
        if have != self.input_dim:
            raise _ir.TraceError("TdnnAffine expects %d input channels, got %d" % (self.input_dim, have))
        w = self.weight.detach().cpu().numpy()
        if self.norm_w:
            # components.py:139-143: F.normalize(weight * mask, dim=1) - every (output channel, tap) column is scaled to unit L2
            # norm over its input channels (eps 1e-12); inactive taps are zero before and after, so the packed active taps get
            # the same numbers.  A constant of the model: folded into the weights here.
            import numpy as np
            wm = w if self.mask is None else w * self.mask.numpy().astype(w.dtype)
            w = (wm / np.maximum(np.sqrt((wm.astype(np.float64) ** 2).sum(axis=1, keepdims=True)), 1e-12)).astype(np.float32)
        if self.groups != 1:
            import numpy as np
            go, gi = self.output_dim // self.groups, self.input_dim // self.groups
            dense = np.zeros((self.output_dim, self.input_dim, w.shape[2]), dtype=w.dtype)
            for g in range(self.groups):
                dense[g * go:(g + 1) * go, g * gi:(g + 1) * gi, :] = w[g * go:(g + 1) * go]
            w = dense
        if order is not None:                  # the pooled tensor's columns are a permutation of the reference's; -1: padding
            import numpy as np
            w = np.where((order >= 0)[None, :, None], w[:, np.maximum(order, 0), :], 0).astype(w.dtype)
        b = self.bias.detach().cpu().numpy() if self.bias is not None else None
        if row_scale is not None:
            import numpy as np
            rs = np.asarray(row_scale, dtype=np.float64)
            w = (w.astype(np.float64) * rs[:, None, None]).astype(np.float32)
            b = None if b is None else (b.astype(np.float64).reshape(-1) * rs).astype(np.float32)
        if row_map is not None:                       # (destination rows, total rows): spread the outputs, zero rows between
            import numpy as np
            dst, total = row_map
            w2 = np.zeros((total,) + w.shape[1:], dtype=w.dtype)
            w2[dst] = w
            b2 = None
            if b is not None:
                b2 = np.zeros(total, dtype=b.dtype)
                b2[dst] = b.reshape(-1)
            w, b = w2, b2
        out = x.graph.tdnn(x.view, w, b, self.context, self.left_context, act1=act1, scale=scale, shift=shift,
                           affine_first=affine_first)
        return _ir.Sym(x.graph, out, 3)

    def forward(self, inputs):
        if isinstance(inputs, _ir.Sym):
            return self.emit(inputs)
        _eager_unsupported("TdnnAffine")

    def extra_repr(self):
        return ("{input_dim}, {output_dim}, context={context}, bias={bool_bias}, stride={stride}, pad={pad}, "
                "groups={groups}, norm_w={norm_w}, norm_f={norm_f}".format(**self.__dict__))


class _BaseActivationBatchNorm(torch.nn.Module):
    """[affine ->] activation -> BatchNorm1d ("bn-relu": BatchNorm1d -> activation)."""

    def __init__(self):
        super(_BaseActivationBatchNorm, self).__init__()
        self.affine = None
        self.activation = None
        self.batchnorm = None
        self.bn_relu = False

    def add_relu_bn(self, output_dim=None, options: dict = {}):
        defaults = {
            "bn-relu": False,
            "nonlinearity": "relu",
            "nonlinearity_params": {"inplace": True, "negative_slope": 0.01},
            "bn": True,
            "ln_replace": False,
            "bn_params": {"momentum": 0.1, "affine": True, "track_running_stats": True},
            "special_init": True,
            "mode": "fan_out",
            "jit_compile": False,
        }
        p = utils.assign_params_dict(defaults, options)
        if p["ln_replace"]:
            raise NotImplementedError("ln_replace (LayerNorm instead of BatchNorm) is not implemented on the MI355X path")
        self.bn_relu = bool(p["bn-relu"])
        # registration order matches the reference so printed models / state_dict order agree
        if not self.bn_relu:
            self.activation = Nonlinearity(p["nonlinearity"], **p["nonlinearity_params"])
            if p["bn"]:
                self.batchnorm = torch.nn.BatchNorm1d(output_dim, **p["bn_params"])
        else:
            if p["bn"]:
                self.batchnorm = torch.nn.BatchNorm1d(output_dim, **p["bn_params"])
            self.activation = Nonlinearity(p["nonlinearity"], **p["nonlinearity_params"])
        if p["special_init"] and self.affine is not None and not p["jit_compile"]:
            if p["nonlinearity"] in ("relu", "leaky_relu", "tanh", "sigmoid"):
                torch.nn.init.kaiming_uniform_(self.affine.weight, a=0, mode=p["mode"], nonlinearity=p["nonlinearity"])
            else:
                torch.nn.init.xavier_normal_(self.affine.weight, gain=1.0)

    def folded_batchnorm(self):
        bn = self.batchnorm
        if bn is None:
            return None, None
        if bn.running_mean is None:
            raise _ir.TraceError("BatchNorm without running statistics cannot run in eval mode")
        g = bn.weight.detach().cpu().numpy() if bn.affine else None
        b = bn.bias.detach().cpu().numpy() if bn.affine else None
        return _ir.fold_batchnorm(bn.running_mean.detach().cpu().numpy(), bn.running_var.detach().cpu().numpy(), g, b, bn.eps)

    def forward(self, inputs):
        if isinstance(inputs, _ir.Sym):
            scale, shift = self.folded_batchnorm()
            return self.affine.emit(inputs, act1=activation_name(self.activation), scale=scale, shift=shift,
                                    affine_first=self.bn_relu)
        _eager_unsupported(type(self).__name__)


class ReluBatchNormTdnnLayer(_BaseActivationBatchNorm):
    """TDNN-ReLU-BN, the 3-fold layer every target model is built from."""

    def __init__(self, input_dim, output_dim, context=[0], affine_type="tdnn", **options):
        super(ReluBatchNormTdnnLayer, self).__init__()
        affine_options = utils.assign_params_dict({"bias": True, "groups": 1, "norm_w": False, "norm_f": False}, options)
        if affine_type != "tdnn":
            raise NotImplementedError("affine_type '%s' (ChunkSeparationAffine) is not implemented on the MI355X path" % affine_type)
        self.affine = TdnnAffine(input_dim, output_dim, context=context, **affine_options)
        self.add_relu_bn(output_dim, options=options)


class FTdnnBlock(torch.nn.Module):
    """Factorised TDNN block (reference components.py:168-212): a bias-free `factor` affine into a bottleneck over the
    left half of the context, an `affine` back out over the right half, ReLU, BatchNorm, and `bypass_scale` times the
    block input added on top.  On the device: two fused TDNN GEMMs (the second one carries ReLU + folded BN) and, when
    there is a bypass, one elementwise pass  identity * bypass_scale + out.  The semi-orthogonal constraint is a
    training-time weight update and has no extraction counterpart."""

    def __init__(self, input_dim, output_dim, bottleneck_dim, context_size=0, bypass_scale=0.66, pad=True):
        super(FTdnnBlock, self).__init__()
        self.input_dim, self.output_dim, self.bottleneck_dim = input_dim, output_dim, bottleneck_dim
        self.context_size, self.bypass_scale, self.pad = context_size, bypass_scale, pad
        left, right = ([-context_size, 0], [0, context_size]) if context_size > 0 else ([0], [0])
        self.factor = TdnnAffine(input_dim, bottleneck_dim, left, pad=pad, bias=False)
        self.affine = TdnnAffine(bottleneck_dim, output_dim, right, pad=pad, bias=True)
        self.relu = torch.nn.ReLU(inplace=True)
        self.bn = torch.nn.BatchNorm1d(output_dim, momentum=0.1, affine=True, track_running_stats=True)

    def forward(self, inputs):
        if not isinstance(inputs, _ir.Sym):
            _eager_unsupported("FTdnnBlock")
        if self.bypass_scale != 0 and self.input_dim != self.output_dim:
            raise _ir.TraceError("FTdnnBlock: a bypass needs equal input and output widths (%d vs %d)" % (self.input_dim, self.output_dim))
        bn = self.bn
        scale, shift = _ir.fold_batchnorm(bn.running_mean.detach().cpu().numpy(), bn.running_var.detach().cpu().numpy(),
                                          bn.weight.detach().cpu().numpy() if bn.affine else None,
                                          bn.bias.detach().cpu().numpy() if bn.affine else None, bn.eps)
        out = self.affine.emit(self.factor.emit(inputs), act1="relu", scale=scale, shift=shift)
        if self.bypass_scale == 0:
            return out
        import numpy as np
        g = inputs.graph
        c = inputs.view.channels
        mixed = g.eltwise(inputs.view, b=out.view, scale=np.full(c, self.bypass_scale, dtype=np.float32), shift=np.zeros(c, dtype=np.float32))
        return _ir.Sym(g, mixed, 3)


class SEBlock(torch.nn.Module):
    """Squeeze-and-excitation over time for [B, C, T] (reference components.py:565-598): mean over the frames of the
    utterance -> TdnnAffine C -> C/ratio -> ReLU -> TdnnAffine -> Sigmoid -> channel scale.  On the device: one mean
    pooling pass, two pooled-domain GEMMs and one elementwise pass with a per-utterance scale (the same ops the
    ECAPA blueprint's own SE_Connect lowers to)."""

    def __init__(self, input_dim, ratio=16, inplace=True):
        super(SEBlock, self).__init__()
        self.input_dim = input_dim
        self.fc_1 = TdnnAffine(input_dim, input_dim // ratio)
        self.relu = torch.nn.ReLU(inplace=inplace)
        self.fc_2 = TdnnAffine(input_dim // ratio, input_dim)
        self.sigmoid = torch.nn.Sigmoid()

    def forward(self, inputs):
        if not isinstance(inputs, _ir.Sym):
            _eager_unsupported("SEBlock")
        if inputs.view.channels != self.input_dim:
            raise _ir.TraceError("SEBlock expects %d channels, got %d" % (self.input_dim, inputs.view.channels))
        g = inputs.graph
        squeezed = _ir.Sym(g, g.pool(inputs.view, stddev=False), 3)
        hidden = self.fc_1.emit(squeezed, act1="relu")
        gate = self.fc_2.emit(hidden, act1="sigmoid")
        return _ir.Sym(g, g.eltwise(inputs.view, seg_scale=gate.view), 3)


class SEBlock_2D(torch.nn.Module):
    """Squeeze-and-excitation over [B, C, F, T] (reference components.py:600-639):
    global average over (F, T) -> Linear C->C/ratio -> ReLU -> Linear -> Sigmoid -> scale.
    Parameter holder; the ResNet planner of libs.nnet.resnet emits its ops."""

    def __init__(self, in_planes, ratio=16, inplace=True):
        super(SEBlock_2D, self).__init__()
        self.avg_pool = torch.nn.AdaptiveAvgPool2d(1)
        self.fc_1 = torch.nn.Linear(in_planes, in_planes // ratio)
        self.relu = torch.nn.ReLU(inplace=inplace)
        self.fc_2 = torch.nn.Linear(in_planes // ratio, in_planes)
        self.sigmoid = torch.nn.Sigmoid()

    def forward(self, inputs):
        _eager_unsupported("SEBlock_2D")


class InputSequenceNormalization(torch.nn.Module):
    """Per-utterance mean (and std) normalisation of the input features over time (reference
    components.py:751-850, used by ResNetXvector(cmvn=True)): x <- (x - mean_t) / max(std_t, 1e-10) with the
    UNBIASED std of torch.std.  On the device: one pooling pass per segment + one elementwise pass."""

    def __init__(self, mean_norm=True, std_norm=True):
        super(InputSequenceNormalization, self).__init__()
        self.mean_norm, self.std_norm = mean_norm, std_norm
        self.eps = 1e-10

    def forward(self, x, lengths=None):
        if not isinstance(x, _ir.Sym):
            _eager_unsupported("InputSequenceNormalization")
        if not (self.mean_norm or self.std_norm):
            return x
        g = x.graph
        # std = max(sqrt(var_unbiased), eps) == sqrt(max(var, eps^2))
        stats = g.pool(x.view, stddev=True, unbiased=2, var_mode=0, eps=self.eps * self.eps)
        out = g.eltwise(x.view, seg_norm=stats, seg_norm_mode=(1 if self.mean_norm else 0) | (2 if self.std_norm else 0))
        return _ir.Sym(g, out, x.rank)
