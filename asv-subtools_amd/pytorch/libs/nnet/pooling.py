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


class AttentionAlphaComponent(torch.nn.Module):
    """Frame weights alpha = softmax over time of [ReLU(first_affine(x))] -> last_affine (reference pooling.py:231-319).
    Built here for the configuration AttentiveStatisticsPooling uses - one head, shared weights (one logit per frame);
    multi-head / split-input / temperature variants raise.  Parameter holder: `logits(x)` emits the affine layer(s)
    as fused TDNN ops (any `context`), the softmax itself runs inside the pooling kernel."""

    def __init__(self, input_dim, num_head=1, split_input=True, share=True, affine_layers=2, hidden_size=64, context=[0], bias=True,
                 temperature=False, fixed=True):
        super(AttentionAlphaComponent, self).__init__()
        if num_head != 1 or not share or temperature:
            raise NotImplementedError("AttentionAlphaComponent(num_head=%d, share=%s, temperature=%s): only the single shared head is built "
                                      "on the MI355X path (SURVEY.md 8(f) rank 3)" % (num_head, share, temperature))
        if affine_layers not in (1, 2):
            raise ValueError("Expected 1 or 2 affine layers, but got {}.".format(affine_layers))
        from .components import TdnnAffine
        self.input_dim, self.num_head, self.share = input_dim, num_head, share
        self.relu_affine = affine_layers == 2
        last_in = input_dim
        if self.relu_affine:
            self.first_affine = TdnnAffine(input_dim, hidden_size, context=context, bias=bias)
            self.relu = torch.nn.ReLU(inplace=True)
            last_in = hidden_size
        self.last_affine = TdnnAffine(last_in, 1, context=context, bias=bias)
        self.softmax = torch.nn.Softmax(dim=2)

    def logits(self, x):
        h = self.first_affine.emit(x, act1="relu") if self.relu_affine else x
        return self.last_affine.emit(h)

    def forward(self, inputs):
        raise NotImplementedError("AttentionAlphaComponent is consumed by AttentiveStatisticsPooling on the MI355X path; it has no stand-alone forward")


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
        if not isinstance(inputs, _ir.Sym):
            raise NotImplementedError("AttentiveStatisticsPooling.forward() on a torch tensor: eager forward is not part of asv-subtools_amd")
        if inputs.view.channels != self.input_dim:
            raise _ir.TraceError("AttentiveStatisticsPooling expects %d channels, got %d" % (self.input_dim, inputs.view.channels))
        g = inputs.graph
        logits = self.attention.logits(inputs)
        both = g.attpool(inputs.view, logits.view, eps=self.eps, shared=True)        # [mean(C) | std(C)]
        if self.stddev:
            return _ir.Sym(g, both, 3)
        return _ir.Sym(g, _ir.View(both.tid, both.ch_off, self.input_dim), 3)       # the mean half of [mean | std]

    def get_output_dim(self):
        return self.output_dim


def _not_on_hot_path(name, where):
    class _Unsupported(torch.nn.Module):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError("%s (%s) is a selectable option of the reference that is outside the "
                                      "MI355X extraction path (SURVEY.md section 2, row 3)" % (name, where))
    _Unsupported.__name__ = name
    return _Unsupported


FreeStatisticsPooling = _not_on_hot_path("FreeStatisticsPooling", "pooling.py:78")
LDEPooling = _not_on_hot_path("LDEPooling", "pooling.py:112")
MultiHeadAttentionPooling = _not_on_hot_path("MultiHeadAttentionPooling", "pooling.py:371")
GlobalMultiHeadAttentionPooling = _not_on_hot_path("GlobalMultiHeadAttentionPooling", "pooling.py:446")
MultiResolutionMultiHeadAttentionPooling = _not_on_hot_path("MultiResolutionMultiHeadAttentionPooling", "pooling.py:513")
MQMHASP = _not_on_hot_path("MQMHASP", "pooling.py:590-701")
