# -*- coding:utf-8 -*-
"""Pooling layers of the extraction path (reference libs/nnet/pooling.py).  StatisticsPooling (15-76) is on the hot
path of the target models; AttentiveStatisticsPooling (322-368, single shared head) is the first of the alternative
poolings of SURVEY.md 8(f) rank 3; the other variants the reference offers are selectable options that this package does
not implement and says so when constructed."""

import torch

from libs.amd import ir as _ir


class StatisticsPooling(torch.nn.Module):
    """mean [+ stddev] over the frames of each utterance: [B, C, T] -> [B, 2C, 1]."""

    def __init__(self, input_dim, stddev=True, unbiased=False, eps=1.0e-10):
        super(StatisticsPooling, self).__init__()
        self.stddev = stddev
        self.input_dim = input_dim
        self.output_dim = 2 * input_dim if stddev else input_dim
        self.eps = eps
        self.unbiased = unbiased

    def forward(self, inputs, lengths=None):
        if isinstance(inputs, _ir.Sym):
            g = inputs.graph.grid_spec(inputs.view.tid)
            if g is not None:
                # ResNetXvector: [B, C*F', T'] with channel index c*F' + f (resnet_xvector.py:193).  The device pools
                # every frequency bin over time and lays the row out bin-major: [f][mean(C) | std(C)]
                if not inputs.flat_grid or inputs.view.channels * g[2] != self.input_dim:
                    raise _ir.TraceError("StatisticsPooling over a 2-D tensor expects the [B, C*F, T] reshape with C*F == %d" % self.input_dim)
                import numpy as np
                C, F = inputs.view.channels, g[2]
                nstat = 2 if self.stddev else 1
                out = inputs.graph.pool(inputs.view, stddev=self.stddev, unbiased=1 if self.unbiased else 0, var_mode=0, eps=self.eps, per_bin=True)
                mine = np.arange(F * nstat * C)
                f, stat, c = mine // (nstat * C), (mine // C) % nstat, mine % C
                return _ir.Sym(inputs.graph, out, 3, col_order=stat * (C * F) + c * F + f)
            if inputs.view.channels != self.input_dim:
                raise _ir.TraceError("StatisticsPooling expects %d channels, got %d" % (self.input_dim, inputs.view.channels))
            # true per-utterance lengths are always used on the HIP path (ragged batches)
            out = inputs.graph.pool(inputs.view, stddev=self.stddev, unbiased=1 if self.unbiased else 0, var_mode=0, eps=self.eps)
            return _ir.Sym(inputs.graph, out, 3)
        raise NotImplementedError("StatisticsPooling.forward() on a torch tensor: eager forward is not part of "
                                  "asv-subtools_amd; use model.extract_embedding(feats)")

    def get_output_dim(self):
        return self.output_dim

    def extra_repr(self):
        return "{input_dim}, {output_dim}, stddev={stddev}, unbiased={unbiased}, eps={eps}".format(**self.__dict__)


class FreeStatisticsPooling(StatisticsPooling):
    """The reference's StatisticsPooling without a declared width (pooling.py:92-127): flattens everything between the batch
    and the frame axis and pools it - the same device op; the width is taken from the tensor it is applied to."""

    def __init__(self, stddev=True, unbiased=False, eps=1.0e-10):
        super(FreeStatisticsPooling, self).__init__(0, stddev=stddev, unbiased=unbiased, eps=eps)

    def forward(self, inputs, lengths=None):
        if isinstance(inputs, _ir.Sym):
            g = inputs.graph.grid_spec(inputs.view.tid)
            self.input_dim = inputs.view.channels * (g[2] if g is not None else 1)
            self.output_dim = 2 * self.input_dim if self.stddev else self.input_dim
            if g is not None and not inputs.flat_grid:
                inputs = inputs.reshape(1, self.input_dim, -1)          # its own reshape(B, -1, T)
        return super(FreeStatisticsPooling, self).forward(inputs, lengths)

    def extra_repr(self):
        return "stddev={stddev}, unbiased={unbiased}, eps={eps}".format(**self.__dict__)


class AttentionAlphaComponent(torch.nn.Module):
    """Frame weights alpha = softmax over time of [ReLU(first_affine(x))] -> last_affine [/ temperature]
    (reference pooling.py:220-319), in all its configurations: one or several heads, heads over channel splits
    (`split_input`) or over all channels, shared (one logit per head and frame) or un-shared (one per channel) last affine,
    one or two affine layers with any `context`, fixed or learned per-head temperatures.  Parameter holder with the
    reference's parameter names and shapes (grouped TdnnAffine weights): `logits(x)` emits the affine layer(s) as fused
    TDNN ops - grouped ones as block-diagonal dense weights, the temperature folded into the last affine - and the
    softmax itself runs inside the pooling kernel."""

    def __init__(self, input_dim, num_head=1, split_input=True, share=True, affine_layers=2, hidden_size=64, context=[0], bias=True,
                 temperature=False, fixed=True):
        super(AttentionAlphaComponent, self).__init__()
        assert num_head >= 1
        if num_head > 1 and split_input:
            assert input_dim % num_head == 0
        if num_head > 1 and temperature:
            if fixed:
                self.register_buffer('t', torch.tensor([[[[max(1, (i // 2) * 5)]] for i in range(num_head)]]))   # int64 [1, H, 1, 1], as the reference builds it
            else:
                self.t = torch.nn.Parameter(torch.zeros(1, num_head, 1, 1))
        if affine_layers not in (1, 2):
            raise ValueError("Expected 1 or 2 affine layers, but got {}.".format(affine_layers))
        from .components import TdnnAffine
        self.input_dim, self.num_head, self.split_input, self.share = input_dim, num_head, split_input, share
        self.temperature, self.fixed = temperature, fixed
        self.final_dim = 1 if share else (input_dim // num_head if split_input else input_dim)
        first_groups, last_groups = 1, 1
        self.relu_affine = affine_layers == 2
        if not self.relu_affine:
            last_in = input_dim
            if num_head > 1 and split_input:
                last_groups = num_head
        else:
            last_in = hidden_size * num_head
            if num_head > 1:
                last_groups = num_head
                if split_input:
                    first_groups = num_head
            self.first_affine = TdnnAffine(input_dim, last_in, context=context, bias=bias, groups=first_groups)
            self.relu = torch.nn.ReLU(inplace=True)
        self.last_affine = TdnnAffine(last_in, self.final_dim * num_head, context=context, bias=bias, groups=last_groups)
        self.softmax = torch.nn.Softmax(dim=2)

    def logits(self, x, head_stride=None):
        """[frames, final_dim * num_head] logits, head-major like the reference's conv output (pooling.py:383-388); with
        `head_stride` every head's final_dim columns start at a multiple of it (device views want 16-channel alignment)."""
        h = self.first_affine.emit(x, act1="relu") if self.relu_affine else x
        row_scale = None
        if self.num_head > 1 and self.temperature:
            t = self.t.detach().cpu().numpy().reshape(-1).astype("float64")
            if not self.fixed:
                t = 1.0 + t * t
            import numpy as np
            row_scale = np.repeat(1.0 / t, self.final_dim)
        row_map = None
        if head_stride is not None and head_stride != self.final_dim:
            import numpy as np
            rows = np.arange(self.final_dim * self.num_head)
            row_map = ((rows // self.final_dim) * head_stride + rows % self.final_dim, head_stride * self.num_head)
        return self.last_affine.emit(h, row_scale=row_scale, row_map=row_map)

    def forward(self, inputs):
        raise NotImplementedError("AttentionAlphaComponent is consumed by the attentive poolings on the MI355X path; it has no stand-alone forward")


def _attentive_stats(inputs, want_dim, what):
    """The traced input of a frame-weighting pooling.  Behind the 2-D trunk (ResNetXvector, resnet_xvector.py:104-111,193) it is
    the [B, C*F', T'] reshape of a grid tensor: materialised here in the reference's channel order (Graph.flatten_grid)."""
    if not isinstance(inputs, _ir.Sym):
        raise NotImplementedError("%s.forward() on a torch tensor: eager forward is not part of asv-subtools_amd" % what)
    inputs = inputs.as_sequence()
    if inputs.view.channels != want_dim:
        raise _ir.TraceError("%s expects %d channels, got %d" % (what, want_dim, inputs.view.channels))
    return inputs


class AttentiveStatisticsPooling(torch.nn.Module):
    """Attention-weighted mean [and std] over the frames of each utterance (reference pooling.py:322-368):
    mean = sum_t alpha_t x_t, std = sqrt(clamp(sum_t alpha_t x_t^2 - mean^2, eps)) with alpha from AttentionAlphaComponent.
    One or two small frame-level GEMMs for the logits, then the softmax-weighted pooling kernel (`shared` logits)."""

    def __init__(self, input_dim, affine_layers=2, hidden_size=64, context=[0], stddev=True, stddev_attention=True, eps=1.0e-10):
        super(AttentiveStatisticsPooling, self).__init__()
        if stddev and not stddev_attention:
            raise NotImplementedError("AttentiveStatisticsPooling(stddev_attention=False) is not built on the MI355X path")
        self.stddev, self.input_dim, self.eps, self.stddev_attention = stddev, input_dim, eps, stddev_attention
        self.output_dim = 2 * input_dim if stddev else input_dim
        self.attention = AttentionAlphaComponent(input_dim, num_head=1, share=True, affine_layers=affine_layers, hidden_size=hidden_size, context=context)

    def forward(self, inputs):
        inputs = _attentive_stats(inputs, self.input_dim, "AttentiveStatisticsPooling")
        g = inputs.graph
        logits = self.attention.logits(inputs)
        both = g.attpool(inputs.view, logits.view, eps=self.eps, shared=True)        # [mean(C) | std(C)]
        if self.stddev:
            return _ir.Sym(g, both, 3)
        return _ir.Sym(g, _ir.View(both.tid, both.ch_off, self.input_dim), 3)       # the mean half of [mean | std]

    def get_output_dim(self):
        return self.output_dim


class MultiHeadAttentionPooling(torch.nn.Module):
    """Heads over channel splits (reference pooling.py:371-438): channel c is weighted by head c // (C / num_head).  Shared
    weights: one logit column per head, the pooling kernel maps channel groups to columns; un-shared: a logit per channel."""

    def __init__(self, input_dim, stddev=True, stddev_attention=True, num_head=4, share=True, affine_layers=1, **options):
        super(MultiHeadAttentionPooling, self).__init__()
        if stddev and not stddev_attention:
            raise NotImplementedError("MultiHeadAttentionPooling(stddev_attention=False) is not built on the MI355X path")
        self.input_dim, self.stddev, self.stddev_attention, self.num_head, self.share = input_dim, stddev, stddev_attention, num_head, share
        self.output_dim = 2 * input_dim if stddev else input_dim
        if "split_input" in options.keys():
            if not options["split_input"]:
                raise ValueError("split_input==False is not valid for this MultiHeadAttentionPooling.")
            options.pop("split_input")
        self.attention = AttentionAlphaComponent(input_dim, num_head=num_head, split_input=True, share=share,
                                                 affine_layers=affine_layers, bias=False, **options)

    def forward(self, inputs):
        inputs = _attentive_stats(inputs, self.input_dim, "MultiHeadAttentionPooling")
        g = inputs.graph
        logits = self.attention.logits(inputs)
        if self.share and self.num_head > 1:
            both = g.attpool(inputs.view, logits.view, eps=1.0e-10, group=self.input_dim // self.num_head)
        else:
            both = g.attpool(inputs.view, logits.view, eps=1.0e-10, shared=self.share)
        if self.stddev:
            return _ir.Sym(g, both, 3)
        return _ir.Sym(g, _ir.View(both.tid, both.ch_off, self.input_dim), 3)

    def get_output_dim(self):
        return self.output_dim


class GlobalMultiHeadAttentionPooling(torch.nn.Module):
    """Every head weights ALL channels (reference pooling.py:441-513): num_head pooling passes over the same frames, head h
    with logit column h (shared) or columns [h C, (h+1) C) (un-shared).  The device row is [mean_1 | std_1 | mean_2 | ...];
    the reference's [mean_1 .. mean_H | std_1 .. std_H] order is restored by permuting the next layer's weight columns."""

    _temperature = False

    def __init__(self, input_dim, stddev=True, stddev_attention=True, num_head=4, share=True, affine_layers=2, **options):
        super(GlobalMultiHeadAttentionPooling, self).__init__()
        if stddev and not stddev_attention:
            raise NotImplementedError("%s(stddev_attention=False) is not built on the MI355X path" % type(self).__name__)
        self.input_dim, self.num_head, self.stddev, self.stddev_attention, self.share = input_dim, num_head, stddev, stddev_attention, share
        self.output_dim = 2 * input_dim if stddev else input_dim
        if "split_input" in options.keys():
            if options["split_input"]:
                raise ValueError("split_input==True is not valid for %s." % type(self).__name__)
            options.pop("split_input")
        if "temperature" in options.keys():
            if bool(options["temperature"]) != self._temperature:
                raise ValueError("temperature==%s is not valid for %s." % (options["temperature"], type(self).__name__))
            options.pop("temperature")
        self.attention = AttentionAlphaComponent(input_dim, num_head=num_head, split_input=False, share=share, temperature=self._temperature,
                                                 affine_layers=affine_layers, bias=True, **options)

    def forward(self, inputs):
        import numpy as np
        inputs = _attentive_stats(inputs, self.input_dim, type(self).__name__)
        g = inputs.graph
        C, H = self.input_dim, self.num_head
        per_head = 1 if self.share else C
        stride = 1 if self.share else -(-C // 16) * 16                # un-shared heads: C logit columns each, 16-aligned starts
        logits = self.attention.logits(inputs, head_stride=stride).view
        parts = [g.attpool(inputs.view, _ir.View(logits.tid, logits.ch_off + h * stride, per_head), eps=1.0e-10, shared=self.share) for h in range(H)]
        if H == 1:
            both = parts[0]
            return _ir.Sym(g, both if self.stddev else _ir.View(both.tid, both.ch_off, C), 3)
        # device row: [mean_h | std_h] blocks at 16-aligned starts; reference row: [mean_1 .. mean_H | std_1 .. std_H] (or the
        # means only).  The next layer's weight columns are gathered accordingly, gaps and unused stds get zero weight.
        block = -(-2 * C // 16) * 16
        out = g.cat(parts, align=16)
        order = np.full(out.channels, -1, dtype=np.int64)
        for h in range(H):
            order[h * block:h * block + C] = h * C + np.arange(C)
            if self.stddev:
                order[h * block + C:h * block + 2 * C] = H * C + h * C + np.arange(C)
        return _ir.Sym(g, out, 3, col_order=order)

    def get_output_dim(self):
        return self.output_dim * self.num_head


class MultiResolutionMultiHeadAttentionPooling(GlobalMultiHeadAttentionPooling):
    """Global heads with per-head softmax temperatures (reference pooling.py:516-587); the temperature is folded into the
    last attention affine."""

    _temperature = True


class xivec_stdinit_softplus2_prec_pooling(torch.nn.Module):
    """xi-vector posterior inference pooling (reference pooling.py:165-218): a per-channel frame precision
    softplus(lin2(ReLU-BN(lin1 x)))^2 weights the frames, a learned Gaussian prior (mean, log-precision) joins as one more
    frame; output = posterior mean [and std].  Two frame-level GEMMs for the precision estimator, then the attentive pooling
    kernel with the 2 log softplus transform and the prior frame."""

    def __init__(self, input_dim, hidden_size=256, context=[0], stddev=False, train_mean=True, train_prec=True):
        super(xivec_stdinit_softplus2_prec_pooling, self).__init__()
        from .components import ReluBatchNormTdnnLayer, TdnnAffine
        self.input_dim, self.stddev = input_dim, stddev
        self.output_dim = 2 * input_dim if stddev else input_dim
        self.prior_mean = torch.nn.Parameter(torch.zeros(1, input_dim), requires_grad=train_mean)
        self.prior_logprec = torch.nn.Parameter(torch.zeros(1, input_dim), requires_grad=train_prec)
        self.softmax = torch.nn.Softmax(dim=2)
        self.lin1_relu_bn = ReluBatchNormTdnnLayer(input_dim, hidden_size, context)
        self.lin2 = TdnnAffine(hidden_size, input_dim, context=context)
        self.softplus2 = torch.nn.Softplus(beta=1, threshold=20)

    def forward(self, inputs):
        inputs = _attentive_stats(inputs, self.input_dim, "xivec_stdinit_softplus2_prec_pooling")
        g = inputs.graph
        prec = self.lin2(self.lin1_relu_bn(inputs))
        both = g.attpool(inputs.view, prec.view, eps=1.0e-10, softplus2=True,
                         prior_logit=self.prior_logprec.detach().cpu().numpy(), prior_value=self.prior_mean.detach().cpu().numpy())
        if self.stddev:
            return _ir.Sym(g, both, 3)
        return _ir.Sym(g, _ir.View(both.tid, both.ch_off, self.input_dim), 3)

    def get_output_dim(self):
        return self.output_dim


class LDEPooling(torch.nn.Module):
    """Learnable dictionary encoding (reference pooling.py:130-162): soft assignment of every frame to c_num centres by its
    scaled squared distance, output = mean residual to each centre, [input_dim * c_num] with column c * c_num + k."""

    def __init__(self, input_dim, c_num=64, eps=1.0e-10):
        super(LDEPooling, self).__init__()
        self.input_dim, self.output_dim, self.eps = input_dim, input_dim * c_num, eps
        self.mu = torch.nn.Parameter(torch.randn(input_dim, c_num))
        self.s = torch.nn.Parameter(torch.ones(c_num))
        self.softmax_for_w = torch.nn.Softmax(dim=3)

    def forward(self, inputs):
        inputs = _attentive_stats(inputs, self.input_dim, "LDEPooling")
        s = self.s.detach().cpu().numpy().astype("float32")
        out = inputs.graph.lde(inputs.view, self.mu.detach().cpu().numpy(), s * s + self.eps)
        return _ir.Sym(inputs.graph, out, 3)

    def get_output_dim(self):
        return self.output_dim


def _not_on_hot_path(name, where):
    class _Unsupported(torch.nn.Module):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError("%s (%s) is a selectable option of the reference that is outside the "
                                      "MI355X extraction path (SURVEY.md section 2, row 3)" % (name, where))
    _Unsupported.__name__ = name
    return _Unsupported


MQMHASP = _not_on_hot_path("MQMHASP", "pooling.py:590-701")
