# -*- coding:utf-8 -*-
"""ResNet x-vector blueprint (ResNet34-SE over 2-D fbank) for the MI355X extraction path.

Public surface of the reference blueprint (/root/reference/pytorch/model/resnet_xvector.py:15-208):
class name, `init` arguments, sub-module names (state_dict keys `resnet.*`, `fc1.*`, `fc2.*`) and the
`extract_embedding` positions, with every pooling the reference's constructor selects (resnet_xvector.py:104-111): statistics
(pooled per frequency bin straight from the grid), and - over the [B, C*F', T'] reshape materialised once on the device -
attentive, multi-head, multi-resolution and LDE.
"""

import sys

import torch

sys.path.insert(0, "subtools/pytorch")

import libs.support.utils as utils
from libs.nnet import *  # noqa: F401,F403


class ResNetXvector(TopVirtualNnet):
    def init(self, inputs_dim, num_targets, aug_dropout=0., tail_dropout=0., training=True, extracted_embedding="near", cmvn=False, cmvn_params={},
             resnet_params={}, pooling="statistics", pooling_params={}, fc1=False, fc1_params={}, fc2_params={}, margin_loss=False,
             margin_loss_params={}, use_step=False, step_params={}, transfer_from="softmax_loss", jit_compile=False):
        resnet_defaults = {
            "head_conv": True, "head_conv_params": {"kernel_size": 3, "stride": 1, "padding": 1},
            "head_maxpool": False, "head_maxpool_params": {"kernel_size": 3, "stride": 1, "padding": 1},
            "block": "BasicBlock", "layers": [3, 4, 6, 3], "planes": [32, 64, 128, 256], "use_se": False, "se_ratio": 4, "convXd": 2,
            "norm_layer_params": {"momentum": 0.5, "affine": True}, "full_pre_activation": True, "zero_init_residual": False,
        }
        fc_defaults = {"nonlinearity": "relu", "nonlinearity_params": {"inplace": True}, "bn-relu": False, "bn": True,
                       "bn_params": {"momentum": 0.5, "affine": True, "track_running_stats": True}}
        resnet_params = utils.assign_params_dict(resnet_defaults, resnet_params)
        pooling_params = utils.assign_params_dict({"num_head": 1, "hidden_size": 64, "share": True, "affine_layers": 1, "context": [0],
                                                   "stddev": True, "temperature": False, "fixed": True}, pooling_params)
        fc1_params = utils.assign_params_dict(fc_defaults, fc1_params)
        fc2_params = utils.assign_params_dict(fc_defaults, fc2_params)
        cmvn_params = utils.assign_params_dict({"mean_norm": True, "std_norm": False}, cmvn_params)

        self.extracted_embedding = extracted_embedding
        self.inputs_dim = inputs_dim
        self.convXd = resnet_params["convXd"]
        self.cmvn_ = InputSequenceNormalization(**cmvn_params) if cmvn else torch.nn.Identity()
        self.resnet = ResNet(1 if self.convXd == 2 else inputs_dim, **resnet_params)
        mult = self.resnet.get_downsample_multiple()
        trunk_dim = (inputs_dim + mult - 1) // mult * self.resnet.get_output_planes()
        stddev = pooling_params.pop("stddev")                            # the reference's selection, resnet_xvector.py:103-115
        if pooling == "lde":
            self.stats = LDEPooling(trunk_dim, c_num=pooling_params["num_head"])
        elif pooling == "attentive":
            self.stats = AttentiveStatisticsPooling(trunk_dim, hidden_size=pooling_params["hidden_size"], context=pooling_params["context"], stddev=stddev)
        elif pooling == "multi-head":
            self.stats = MultiHeadAttentionPooling(trunk_dim, stddev=stddev, **pooling_params)
        elif pooling == "multi-resolution":
            self.stats = MultiResolutionMultiHeadAttentionPooling(trunk_dim, **pooling_params)
        else:
            self.stats = StatisticsPooling(trunk_dim, stddev=stddev)
        embd = resnet_params["planes"][3]
        self.fc1 = ReluBatchNormTdnnLayer(self.stats.get_output_dim(), embd, **fc1_params) if fc1 else None
        self.fc2 = ReluBatchNormTdnnLayer(embd if fc1 else self.stats.get_output_dim(), embd, **fc2_params)
        self.embd_dim = embd
        if training:
            self.loss = MarginSoftmaxLoss(embd, num_targets, **margin_loss_params) if margin_loss else SoftmaxLoss(embd, num_targets)

    @for_extract_embedding(maxChunk=10000, isMatrix=True)
    def extract_embedding(self, x):
        x = self.cmvn_(x)
        x = x.unsqueeze(1)                                            # [B, F, T] -> [B, 1, F, T]
        x = self.resnet(x)
        x = x.reshape(x.shape[0], x.shape[1] * x.shape[2], x.shape[3])  # channel index c*F' + f
        x = self.stats(x)
        if self.extracted_embedding == "far":
            assert self.fc1 is not None
            return self.fc1.affine(x)
        x = self.auto(self.fc1, x)
        if self.extracted_embedding == "near_affine":
            return self.fc2.affine(x)
        if self.extracted_embedding == "near":
            return self.fc2(x)
        raise TypeError("Expected far or near position, but got {}".format(self.extracted_embedding))
