# -*- coding:utf-8 -*-
"""Factorised TDNN (TDNN-F) x-vector blueprint for the MI355X extraction path - SURVEY.md section 8(f) rank 3.

Constructor signature, sub-module names (=> state_dict keys) and the two `extract_embedding` positions follow the
reference blueprint (/root/reference/pytorch/model/factored_xvector.py:14-127); that file also traces unmodified
against this package's `libs.nnet` (FTdnnBlock lives in libs/nnet/components.py).  The dense skip wiring
(layer07 sees [x2 ; x4], layer09 sees [x4 ; x6 ; x8]) is concatenation by channel slices of one wide buffer:
producers write into their slice, nothing is copied.
"""

import sys

sys.path.insert(0, "subtools/pytorch")

from libs.nnet import *  # noqa: F401,F403

# layerNN: (input width, context_size, bypass_scale, inputs = indices of earlier layer outputs concatenated)
_FTDNN = {
    2: (512, 2, 0, (1,)),
    3: (1024, 0, 0.66, (2,)),
    4: (1024, 3, 0.66, (3,)),
    5: (1024, 0, 0.66, (3,)),
    6: (1024, 3, 0.66, (5,)),
    7: (2048, 3, 0, (2, 4)),
    8: (1024, 3, 0.66, (7,)),
    9: (3072, 0, 0, (4, 6, 8)),
}


class Xvector(TopVirtualNnet):
    """layer01 (TDNN) -> eight factorised blocks with dense skips -> layer10 -> statistics pooling -> embedding1 ("far")
    -> embedding2 ("near")."""

    def init(self, inputs_dim, num_targets, nonlinearity="relu", semi_orth=True, embd_dim=512, aug_dropout=0.2, training=False,
             extracted_embedding="far", jit_compile=False):
        if training:
            raise NotImplementedError("this blueprint is the extraction graph only (training=False)")
        self.extracted_embedding = extracted_embedding
        self.inputs_dim = inputs_dim
        self.embd_dim = embd_dim
        self.layer01 = ReluBatchNormTdnnLayer(inputs_dim, 512, [-2, -1, 0, 1, 2], nonlinearity=nonlinearity)
        for n in sorted(_FTDNN):
            width, context_size, bypass, _ = _FTDNN[n]
            setattr(self, "layer%02d" % n, FTdnnBlock(width, 1024, 256, context_size, bypass))
        self.layer10 = ReluBatchNormTdnnLayer(1024, 2048, nonlinearity=nonlinearity)
        self.stats = StatisticsPooling(2048, stddev=True)
        self.embedding1 = ReluBatchNormTdnnLayer(self.stats.get_output_dim(), embd_dim, nonlinearity=nonlinearity)
        self.embedding2 = ReluBatchNormTdnnLayer(embd_dim, embd_dim, nonlinearity=nonlinearity)

    @for_extract_embedding(maxChunk=10000, isMatrix=True)
    def extract_embedding(self, inputs):
        import torch
        outs = {1: self.layer01(inputs)}
        for n in sorted(_FTDNN):
            sources = [outs[k] for k in _FTDNN[n][3]]
            outs[n] = getattr(self, "layer%02d" % n)(sources[0] if len(sources) == 1 else torch.cat(sources, 1))
        x = self.stats(self.layer10(outs[9]))
        if self.extracted_embedding == "far":
            return self.embedding1.affine(x)
        if self.extracted_embedding == "near":
            return self.embedding2.affine(self.embedding1(x))
        raise TypeError("Expected far or near position, but got {}".format(self.extracted_embedding))
