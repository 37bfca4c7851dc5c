# -*- coding:utf-8 -*-
"""Standard TDNN x-vector blueprint for the MI355X extraction path.

Same class name, constructor signature, sub-module names (=> same state_dict keys) and
`extract_embedding` positions as the reference blueprint
(/root/reference/pytorch/model/xvector.py:15-98), so `nnet.config` files and `*.params`
checkpoints of the reference work unchanged.  The reference's own xvector.py also runs
unmodified against this package's `libs.nnet`; this copy exists because the reference tree
is not shipped with this repository.
"""

import sys

sys.path.insert(0, "subtools/pytorch")

from libs.nnet import *  # noqa: F401,F403

# (name, out_dim, context) of the five frame-level layers
_FRAME_LAYERS = (
    ("tdnn1", 512, [-2, -1, 0, 1, 2]),
    ("tdnn2", 512, [-2, 0, 2]),
    ("tdnn3", 512, [-3, 0, 3]),
    ("tdnn4", 512, [0]),
    ("tdnn5", 1500, [0]),
)


class Xvector(TopVirtualNnet):
    """tdnn1-5 -> mean/std pooling -> tdnn6 ("far" = its affine) -> tdnn7 ("near" = its affine)."""

    def init(self, inputs_dim, num_targets, nonlinearity="relu", aug_dropout=0.2, training=True, extracted_embedding="far"):
        self.extracted_embedding = extracted_embedding
        self.inputs_dim = inputs_dim
        dim = inputs_dim
        for name, out_dim, context in _FRAME_LAYERS:
            setattr(self, name, ReluBatchNormTdnnLayer(dim, out_dim, context, nonlinearity=nonlinearity))
            dim = out_dim
        self.stats = StatisticsPooling(dim, stddev=True)
        self.tdnn6 = ReluBatchNormTdnnLayer(self.stats.get_output_dim(), 512, nonlinearity=nonlinearity)
        self.tdnn7 = ReluBatchNormTdnnLayer(512, 512, nonlinearity=nonlinearity)
        if training:
            self.loss = SoftmaxLoss(512, num_targets)

    @for_extract_embedding(maxChunk=10000, isMatrix=True)
    def extract_embedding(self, inputs):
        x = inputs
        for name, _, _ in _FRAME_LAYERS:
            x = getattr(self, name)(x)
        x = self.stats(x)
        if self.extracted_embedding == "far":
            return self.tdnn6.affine(x)
        if self.extracted_embedding == "near":
            return self.tdnn7.affine(self.tdnn6(x))
        raise TypeError("Expected far or near position, but got {}".format(self.extracted_embedding))
