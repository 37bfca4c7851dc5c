# -*- coding:utf-8 -*-
"""Pooling layers of the extraction path (reference libs/nnet/pooling.py).  Only
StatisticsPooling (15-76) is on the hot path of the three target models; the other pooling
variants the reference offers are selectable options that this package does not implement
and says so when constructed."""

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


def _not_on_hot_path(name, where):
    class _Unsupported(torch.nn.Module):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError("%s (%s) is a selectable option of the reference that is outside the "
                                      "MI355X extraction path (SURVEY.md section 2, row 3)" % (name, where))
    _Unsupported.__name__ = name
    return _Unsupported


FreeStatisticsPooling = _not_on_hot_path("FreeStatisticsPooling", "pooling.py:78")
LDEPooling = _not_on_hot_path("LDEPooling", "pooling.py:112")
AttentiveStatisticsPooling = _not_on_hot_path("AttentiveStatisticsPooling", "pooling.py:322-368")
MultiHeadAttentionPooling = _not_on_hot_path("MultiHeadAttentionPooling", "pooling.py:371")
GlobalMultiHeadAttentionPooling = _not_on_hot_path("GlobalMultiHeadAttentionPooling", "pooling.py:446")
MultiResolutionMultiHeadAttentionPooling = _not_on_hot_path("MultiResolutionMultiHeadAttentionPooling", "pooling.py:513")
MQMHASP = _not_on_hot_path("MQMHASP", "pooling.py:590-701")
