# -*- coding:utf-8 -*-
"""ECAPA-TDNN blueprint for the MI355X extraction path.

Mirrors the reference blueprint's public surface - class names, `ECAPA_TDNN.init` arguments,
sub-module names (hence state_dict keys), `extract_embedding` positions and
`extract_embedding_whole` / `embedding_dim` (reference
/root/reference/pytorch/model/ecapa_tdnn_xvector.py: Res2NetBlock 17-75, SE_Connect 97-111,
SE_Res2Block 118-149, AttentiveStatsPool 156-188, ECAPA_TDNN 200-482) - so reference
`nnet.config` / `*.params` files work unchanged.  Poolings: the default "ecpa-attentive", "attentive"
(AttentiveStatisticsPooling, reference :275-281) and plain statistics pooling.  "multi-head", "global-multi"
and "multi-resolution" cannot be constructed in the reference either (its pooling defaults hand
`time_attention` to AttentionAlphaComponent: TypeError) and "mqmha" is out of scope: they raise.

All modules are parameter holders whose forward() records fused ops for libasv_amd.so.
"""

import sys

import torch
import torch.nn as nn

sys.path.insert(0, "subtools/pytorch")

import libs.support.utils as utils
from libs.nnet import *  # noqa: F401,F403
from libs.amd import ir as _ir


def _only_symbolic(x, who):
    if not isinstance(x, _ir.Sym):
        raise NotImplementedError("%s.forward() on a torch tensor: eager forward is not part of asv-subtools_amd" % who)


class Res2NetBlock(nn.Module):
    """`scale` channel groups; group 0 passes through, group i feeds TDNN[-d,0,d]+ReLU+BN with
    the previous group's output added (hierarchical residual)."""

    def __init__(self, in_channels, out_channels, scale=8, kernel_size=3, dilation=1, bn_params={}):
        super(Res2NetBlock, self).__init__()
        assert scale > 1 and in_channels % scale == 0 and out_channels % scale == 0
        half = kernel_size // 2
        context = list(range(-half * dilation, half * dilation + 1, dilation))
        self.blocks = nn.ModuleList([ReluBatchNormTdnnLayer(in_channels // scale, out_channels // scale, context, **bn_params)
                                     for _ in range(scale - 1)])
        self.scale = scale

    def forward(self, x):
        _only_symbolic(x, "Res2NetBlock")
        groups = x.chunk(self.scale, dim=1)
        outs, prev = [groups[0]], None
        for i, block in enumerate(self.blocks):
            prev = block(groups[i + 1] if prev is None else prev + groups[i + 1])
            outs.append(prev)
        return _ir.sym_cat(outs, dim=1)


class SE_Connect(nn.Module):
    """Squeeze-excitation: time mean -> 1x1 conv -> ReLU -> 1x1 conv -> sigmoid -> channel scale."""
    _asv_amd_native = True

    def __init__(self, channels, bottleneck=128):
        super(SE_Connect, self).__init__()
        self.se = nn.Sequential(nn.AdaptiveAvgPool1d(1), nn.Conv1d(channels, bottleneck, kernel_size=1, padding=0), nn.ReLU(),
                                nn.Conv1d(bottleneck, channels, kernel_size=1, padding=0), nn.Sigmoid())

    def forward(self, x):
        _only_symbolic(x, "SE_Connect")
        return _ir.MODULE_HANDLERS["SE_Connect"](self, x)


class SE_Res2Block(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, dilation=1, scale=8, bn_params={}):
        super(SE_Res2Block, self).__init__()
        width = (in_channels // scale) * scale
        self.conv_relu_bn1 = ReluBatchNormTdnnLayer(in_channels, width, **bn_params)
        self.res2net_block = Res2NetBlock(width, width, scale=scale, kernel_size=kernel_size, dilation=dilation, bn_params=bn_params)
        self.conv_relu_bn2 = ReluBatchNormTdnnLayer(in_channels, width, **bn_params)
        self.se = SE_Connect(out_channels)
        self.shortcut = nn.Conv1d(in_channels, out_channels, kernel_size=1) if in_channels != out_channels else None

    def forward(self, x):
        _only_symbolic(x, "SE_Res2Block")
        residual = x if self.shortcut is None else self.shortcut(x)
        y = self.se(self.conv_relu_bn2(self.res2net_block(self.conv_relu_bn1(x))))
        return y + residual


class AttentiveStatsPool(nn.Module):
    """Channel- and context-dependent attentive mean/std pooling (ECAPA)."""
    _asv_amd_native = True

    def __init__(self, in_dim, bottleneck_dim=128, time_attention=False, bn={}):
        super(AttentiveStatsPool, self).__init__()
        self.time_attention = time_attention
        accept = in_dim * 3 if time_attention else in_dim
        self.attention = nn.Sequential(nn.Conv1d(accept, bottleneck_dim, kernel_size=1), nn.ReLU(), nn.BatchNorm1d(bottleneck_dim, **bn),
                                       nn.Tanh(), nn.Conv1d(bottleneck_dim, in_dim, kernel_size=1), nn.Softmax(dim=2))

    def forward(self, x):
        _only_symbolic(x, "AttentiveStatsPool")
        return _ir.MODULE_HANDLERS["AttentiveStatsPool"](self, x)


class ECAPA_TDNN(TopVirtualNnet):
    def init(self, inputs_dim, num_targets, aug_dropout=0., tail_dropout=0., training=True, extracted_embedding="near", mixup=False,
             mixup_alpha=1.0, pooling="ecpa-attentive", pooling_params={}, ecapa_params={}, fc1=False, fc1_params={}, fc2_params={},
             margin_loss=True, margin_loss_params={}, use_step=False, step_params={}, transfer_from="softmax_loss"):
        bn_half = {"momentum": 0.5, "affine": True, "track_running_stats": True}
        ecapa_params = utils.assign_params_dict({"channels": 1024, "embd_dim": 192, "mfa_conv": 1536, "bn_params": bn_half}, ecapa_params)
        pooling_params = utils.assign_params_dict({"hidden_size": 128, "time_attention": True, "stddev": True}, pooling_params, support_unknow=True)
        fc_defaults = {"nonlinearity": "relu", "nonlinearity_params": {"inplace": True}, "bn-relu": False, "bn": True, "bn_params": bn_half}
        fc1_params = utils.assign_params_dict(fc_defaults, fc1_params)
        fc2_params = utils.assign_params_dict(fc_defaults, fc2_params)

        self.use_step, self.step_params = use_step, step_params
        self.extracted_embedding = extracted_embedding
        self.inputs_dim = inputs_dim
        self.embd_dim = embd_dim = ecapa_params["embd_dim"]
        C, mfa_dim = ecapa_params["channels"], ecapa_params["mfa_conv"]

        # every layer receives **ecapa_params; only its "bn_params" entry is honoured (assign_params_dict drops unknown keys)
        self.layer1 = ReluBatchNormTdnnLayer(inputs_dim, C, [-2, -1, 0, 1, 2], **ecapa_params)
        self.layer2 = SE_Res2Block(C, C, kernel_size=3, dilation=2, scale=8, bn_params=ecapa_params)
        self.layer3 = SE_Res2Block(C, C, kernel_size=3, dilation=3, scale=8, bn_params=ecapa_params)
        self.layer4 = SE_Res2Block(C, C, kernel_size=3, dilation=4, scale=8, bn_params=ecapa_params)
        self.mfa = ReluBatchNormTdnnLayer(3 * C, mfa_dim, **ecapa_params)

        stddev = pooling_params.pop("stddev")
        if pooling == "ecpa-attentive":
            self.stats = AttentiveStatsPool(mfa_dim, pooling_params["hidden_size"], pooling_params["time_attention"])
            self.bn_stats = nn.BatchNorm1d(mfa_dim * 2, **ecapa_params["bn_params"])
        elif pooling == "attentive":                                   # reference :275-281 (needs pooling_params["context"], like there)
            self.stats = AttentiveStatisticsPooling(mfa_dim, hidden_size=pooling_params["hidden_size"], context=pooling_params["context"], stddev=stddev)
            self.bn_stats = nn.BatchNorm1d(mfa_dim * 2, **ecapa_params["bn_params"])
        elif pooling in ("multi-head", "global-multi", "multi-resolution"):
            raise TypeError("pooling='%s': the reference's ECAPA_TDNN cannot build this option either - its pooling defaults pass "
                            "`time_attention` on to AttentionAlphaComponent, which does not take it (ecapa_tdnn_xvector.py:213-217, 296-316)" % pooling)
        elif pooling == "mqmha":
            raise NotImplementedError("pooling='mqmha' (MQMHASP) is a selectable option of the reference outside the MI355X extraction path "
                                      "(SURVEY.md section 2, row 3)")
        else:
            self.stats = StatisticsPooling(mfa_dim, stddev=stddev)
            self.bn_stats = nn.BatchNorm1d(mfa_dim * 2)
        self.fc1 = ReluBatchNormTdnnLayer(mfa_dim * 2, embd_dim, **fc1_params) if fc1 else None
        self.fc2 = ReluBatchNormTdnnLayer(embd_dim if fc1 else mfa_dim * 2, embd_dim, **fc2_params)
        self.tail_dropout = None
        if training:
            self.loss = MarginSoftmaxLoss_v1(embd_dim, num_targets, **margin_loss_params) if margin_loss else SoftmaxLoss(embd_dim, num_targets)

    def _embed(self, x, position):
        x = self.layer1(x)
        x1 = self.layer2(x)
        x2 = self.layer3(x + x1)
        x3 = self.layer4(x + x1 + x2)
        x = self.mfa(torch.cat([x1, x2, x3], dim=1))
        x = self.bn_stats(self.stats(x))
        if len(x.shape) != 3:
            x = x.unsqueeze(dim=2)
        if position == "far":
            assert self.fc1 is not None
            return self.fc1.affine(x)
        x = self.auto(self.fc1, x)
        if position == "near_affine":
            return self.fc2.affine(x)
        if position == "near":
            return self.fc2(x)
        raise TypeError("Expected far or near position, but got {}".format(position))

    @for_extract_embedding(maxChunk=10000, isMatrix=True)
    def extract_embedding(self, x):
        return self._embed(x, self.extracted_embedding)

    def extract_embedding_whole(self, input, position="near", maxChunk=4000, isMatrix=True):
        """The TorchScript-exported twin the online extractor / C++ runtime call
        (reference ecapa_tdnn_xvector.py:454-476): same chunk rule, maxChunk 4000."""
        from libs.nnet.framework import _to_frames_matrix
        body = _POSITION_BODIES.setdefault(position, lambda self, x, _p=position: self._embed(x, _p))
        emb = self._amd_engine(body).extract_batch([_to_frames_matrix(input, isMatrix)], max_chunk=maxChunk)
        return emb[0]

    def embedding_dim(self):
        return self.embd_dim


_POSITION_BODIES = {}
