# -*- coding:utf-8 -*-
"""Composite TDNN x-vector blueprint ("snowdar" x-vector: optional extension layers, squeeze-excitation blocks, skip
connection, optional tdnn6) for the MI355X extraction path - SURVEY.md section 8(f) rank 3.

Constructor signature, sub-module names (=> state_dict keys) and the three `extract_embedding` positions follow the
reference blueprint (/root/reference/pytorch/model/snowdar_xvector.py:13-278); that file also traces unmodified against
this package's `libs.nnet`.  What is an extraction-time no-op there (mixup, SpecAugment, dropouts, margin / step
parameters) is accepted and ignored.  Every pooling the reference's constructor selects is built: statistics, attentive,
multi-head, multi-resolution, LDE and the two xi-vector poolings (libs/nnet/pooling.py).
"""

import sys

sys.path.insert(0, "subtools/pytorch")

import libs.support.utils as utils
from libs.nnet import *  # noqa: F401,F403

_TDNN_DEFAULTS = {"nonlinearity": "relu", "nonlinearity_params": {"inplace": True}, "bn-relu": False, "bn": True,
                  "bn_params": {"momentum": 0.5, "affine": False, "track_running_stats": True}}
_POOLING_DEFAULTS = {"num_nodes": 1500, "num_head": 1, "share": True, "affine_layers": 1, "hidden_size": 64, "context": [0],
                     "stddev": True, "temperature": False, "fixed": True}

# frame-level stack in execution order: ("tdnn" | "se", attribute, context, needs_extend)
_STACK = (
    ("tdnn", "tdnn1", [-2, -1, 0, 1, 2], False), ("se", "se1", None, False), ("tdnn", "ex_tdnn1", [0], True),
    ("tdnn", "tdnn2", [-2, 0, 2], False), ("se", "se2", None, False), ("tdnn", "ex_tdnn2", [0], True),
    ("tdnn", "tdnn3", [-3, 0, 3], False), ("se", "se3", None, False), ("tdnn", "ex_tdnn3", [0], True),
    ("tdnn", "ex_tdnn4", [-4, 0, 4], True), ("se", "se4", None, True), ("tdnn", "ex_tdnn5", [0], True),
    ("tdnn", "tdnn4", [0], False),
)


class Xvector(TopVirtualNnet):
    """tdnn1 [se1] [ex1] tdnn2 [se2] [ex2] tdnn3 [se3] [ex3] [ex4 [se4] ex5] tdnn4 (+ tdnn1's output if skip_connection)
    tdnn5 -> statistics pooling -> [tdnn6] -> tdnn7."""

    def init(self, inputs_dim, num_targets, extend=False, skip_connection=False, mixup=False, mixup_alpha=1.0, specaugment=False,
             specaugment_params={}, aug_dropout=0., context_dropout=0., hidden_dropout=0., dropout_params={}, SE=False, se_ratio=4,
             tdnn_layer_params={}, tdnn6=True, tdnn7_params={}, pooling="statistics", pooling_params={}, margin_loss=False,
             margin_loss_params={}, use_step=False, step_params={}, transfer_from="softmax_loss", training=True, extracted_embedding="far"):
        layer_params = utils.assign_params_dict(_TDNN_DEFAULTS, tdnn_layer_params)
        last_params = utils.assign_params_dict(layer_params, tdnn7_params)
        pool_params = utils.assign_params_dict(_POOLING_DEFAULTS, pooling_params)
        if training:
            raise NotImplementedError("this blueprint is the extraction graph only (training=False)")
        self.extracted_embedding = extracted_embedding
        self.inputs_dim = inputs_dim
        self.skip_connection = skip_connection
        dim = inputs_dim
        for kind, name, context, needs_extend in _STACK:
            present = (extend or not needs_extend) and (kind == "tdnn" or SE)
            if not present:
                setattr(self, name, None)
            elif kind == "tdnn":
                setattr(self, name, ReluBatchNormTdnnLayer(dim, 512, context, **layer_params))
                dim = 512
            else:
                setattr(self, name, SEBlock(dim, ratio=se_ratio))
        self.tdnn5 = ReluBatchNormTdnnLayer(512, pool_params["num_nodes"], **layer_params)
        head_params = {k: v for k, v in pool_params.items() if k not in ("num_nodes", "stddev")}       # reference :116-130
        if pooling == "lde":
            self.stats = LDEPooling(pool_params["num_nodes"], c_num=pool_params["num_head"])
        elif pooling == "attentive":
            self.stats = AttentiveStatisticsPooling(pool_params["num_nodes"], affine_layers=pool_params["affine_layers"], hidden_size=pool_params["hidden_size"],
                                                    context=pool_params["context"], stddev=pool_params["stddev"])
        elif pooling == "multi-head":
            self.stats = MultiHeadAttentionPooling(pool_params["num_nodes"], stddev=pool_params["stddev"], **head_params)
        elif pooling in ("xi-postmean-softplus2", "xi-postdist-softplus2"):
            self.stats = xivec_stdinit_softplus2_prec_pooling(pool_params["num_nodes"], hidden_size=pool_params["hidden_size"],
                                                              stddev=pooling == "xi-postdist-softplus2")
        elif pooling == "multi-resolution":
            self.stats = MultiResolutionMultiHeadAttentionPooling(pool_params["num_nodes"], **head_params)
        else:
            self.stats = StatisticsPooling(pool_params["num_nodes"], stddev=pool_params["stddev"])
        stats_dim = self.stats.get_output_dim()
        self.tdnn6 = ReluBatchNormTdnnLayer(stats_dim, 512, **layer_params) if tdnn6 else None
        if last_params["nonlinearity"] == "default":
            last_params["nonlinearity"] = layer_params["nonlinearity"]
        self.tdnn7 = ReluBatchNormTdnnLayer(512 if tdnn6 else stats_dim, 512, **last_params)

    @for_extract_embedding(maxChunk=10000, isMatrix=True)
    def extract_embedding(self, inputs):
        x, first = inputs, None
        for _, name, _, _ in _STACK:
            layer = getattr(self, name)
            if layer is not None:
                x = layer(x)
            if name == "tdnn1":
                first = x
        if self.skip_connection:
            x = x + first
        x = self.stats(self.tdnn5(x))
        if self.extracted_embedding == "far":
            if self.tdnn6 is None:
                raise TypeError("the far position is the affine of tdnn6, which this model was built without")
            return self.tdnn6.affine(x)
        if self.tdnn6 is not None:
            x = self.tdnn6(x)
        if self.extracted_embedding == "near_affine":
            return self.tdnn7.affine(x)
        if self.extracted_embedding == "near":
            return self.tdnn7(x)
        raise TypeError("Expected far, near_affine or near position, but got {}".format(self.extracted_embedding))
