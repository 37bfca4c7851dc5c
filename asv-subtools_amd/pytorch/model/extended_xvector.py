# -*- coding:utf-8 -*-
"""Extended TDNN x-vector blueprint (E-TDNN: a 1x1 layer behind every context layer plus a fourth context
layer) for the MI355X extraction path - SURVEY.md section 8(f) rank 3.

Class name, constructor signature, sub-module names (=> state_dict keys) and the two `extract_embedding` positions
are those of the reference blueprint (/root/reference/pytorch/model/extended_xvector.py:13-121), so its
`nnet.config` / `*.params` files load unchanged; the reference's own file also traces unmodified against this
package's `libs.nnet`.  Every layer is the same fused TDNN op as in xvector.py: ten frame-level GEMM launches
(the last one with the statistics pooling folded into its epilogue), then the pooled affine.
"""

import sys

sys.path.insert(0, "subtools/pytorch")

from libs.nnet import *  # noqa: F401,F403

# frame-level stack in execution order: (attribute, out_dim, context, only_when_extended)
_STACK = (
    ("tdnn1", 512, [-2, -1, 0, 1, 2], False),
    ("ex_tdnn1", 512, [0], True),
    ("tdnn2", 512, [-2, 0, 2], False),
    ("ex_tdnn2", 512, [0], True),
    ("tdnn3", 512, [-3, 0, 3], False),
    ("ex_tdnn3", 512, [0], True),
    ("ex_tdnn4", 512, [-4, 0, 4], True),
    ("ex_tdnn5", 512, [0], True),
    ("tdnn4", 512, [0], False),
    ("tdnn5", 1500, [0], False),
)
# registration order of the reference (state_dict order / printed model): tdnnN and ex_tdnnN interleaved, then 4 and 5
_REGISTRATION = ("tdnn1", "ex_tdnn1", "tdnn2", "ex_tdnn2", "tdnn3", "ex_tdnn3", "ex_tdnn4", "ex_tdnn5", "tdnn4", "tdnn5")


class ExtendedXvector(TopVirtualNnet):
    """E-TDNN -> mean/std pooling -> tdnn6 ("far" = its affine) -> tdnn7 ("near" = its affine)."""

    def init(self, inputs_dim, num_targets, extend=True, nonlinearity="relu", aug_dropout=0.2, training=True, extracted_embedding="far"):
        self.extracted_embedding = extracted_embedding
        self.inputs_dim = inputs_dim
        spec = {name: (out_dim, context, ext) for name, out_dim, context, ext in _STACK}
        dims, dim = {}, inputs_dim
        for name, out_dim, _, ext in _STACK:                       # input width of every layer that exists
            if ext and not extend:
                continue
            dims[name], dim = dim, out_dim
        for name in _REGISTRATION:
            out_dim, context, ext = spec[name]
            setattr(self, name, ReluBatchNormTdnnLayer(dims[name], out_dim, context, nonlinearity=nonlinearity) if name in dims else None)
        self.stats = StatisticsPooling(dim, stddev=True)
        self.tdnn6 = ReluBatchNormTdnnLayer(self.stats.get_output_dim(), 512, nonlinearity=nonlinearity)
        self.tdnn7 = ReluBatchNormTdnnLayer(512, 512, nonlinearity=nonlinearity)
        if training:
            self.loss = SoftmaxLoss(512, num_targets)

    @for_extract_embedding(maxChunk=10000, isMatrix=True)
    def extract_embedding(self, inputs):
        x = inputs
        for name, _, _, _ in _STACK:
            layer = getattr(self, name)
            if layer is not None:
                x = layer(x)
        x = self.stats(x)
        if self.extracted_embedding == "far":
            return self.tdnn6.affine(x)
        if self.extracted_embedding == "near":
            return self.tdnn7.affine(self.tdnn6(x))
        raise TypeError("Expected far or near position, but got {}".format(self.extracted_embedding))
