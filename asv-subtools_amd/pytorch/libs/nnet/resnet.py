# -*- coding:utf-8 -*-
"""2-D ResNet trunk of ResNetXvector: same class names, constructor arguments and state_dict keys
as the reference (/root/reference/pytorch/libs/nnet/resnet.py: conv3x3/conv1x1 12-20, BasicBlock
23-110, ResNet 212-371), parameter holders whose forward() records the trunk for libasv_amd.so.

Mapping to the device program: a [B, C, F, T] tensor is a "grid" buffer whose
rows are (time, frequency) positions, frequency fastest, with one zero row after the F bins of every
frame.  A stride-1 3x3 convolution is then a 9-tap layer with row offsets dt*(F+1) + df - the same
implicit-GEMM kernel as the TDNN layers with a wider staged window; stride-2 convolutions (first block
of layers 2-4 and their 1x1 down-sampling branch) go through an im2col gather + one GEMM.  Eval
BatchNorm follows the convolution directly, so it is folded into the weights (scale) and bias (shift).

Implemented: BasicBlock in both forms - the original conv-BN-ReLU one (BASELINE config C5 and the reference launchers,
runResnetXvector_online.py:221-260) and the full pre-activation BN-ReLU-conv one (the blueprint's default,
resnet_xvector.py:38; resnet.py:59-104) - with optional SE, head conv 3x3 / stride 1, no max-pool.  Other options raise.
"""

import numpy as np
import torch
import torch.nn as nn

from libs.amd import ir as _ir
from libs.nnet.components import SEBlock_2D


def conv3x3(in_planes, out_planes, Conv=nn.Conv2d, stride=1, groups=1, dilation=1):
    return Conv(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups, bias=False, dilation=dilation)


def conv1x1(in_planes, out_planes, Conv=nn.Conv2d, stride=1):
    return Conv(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


def _np(t):
    return t.detach().cpu().numpy()


def _folded_bn(bn):
    g = _np(bn.weight) if bn.affine else None
    b = _np(bn.bias) if bn.affine else None
    return _ir.fold_batchnorm(_np(bn.running_mean), _np(bn.running_var), g, b, bn.eps)


# Stride-2 convolutions with this many input channels are lowered as space-to-depth gather + 2 x 2 convolution on the output
# grid (see emit_conv_bn).  Measured on the ResNet34 trunk (r2n, 256 x 200 frames): 64 -> 128: gather 181 + 22 -> 84 us, convolution
# 97 -> 123 us (its 256 -> 128 4-tap form runs on the generic 128 x 128 tile): -93 us; 32 -> 64: gather 344 + 46 -> 157 us but
# the 128 -> 64 4-tap form falls to 87 TFLOP/s on that tile (233 -> 436 us): taken since round 3, with a kernel of its own for that
# form (kernels_conv2d.hip grid_conv_s2d_kernel; profiles/r3l_*); 128 -> 256: the 512-channel form would not beat im2col + the
# 256-channel tiles: not taken.
S2D_CIN = (32, 64)


def s2d_kernel(w, pitch_out):
    """3 x 3 stride-2 kernel w [Cout, Cin, kF, kT] -> (taps, left, dense [Cout, 4 Cin, taps[-1] - left + 1]) of the equivalent
    2 x 2 stride-1 convolution over the space-to-depth tensor [rows'][(pt * 2 + pf) * Cin + c] on the output grid of pitch
    `pitch_out` (see emit_conv_bn): input offset dt in {-1, 0, 1} is phase |dt| of output-row offset (dt < 0 ? -1 : 0)."""
    cout, cin = w.shape[0], w.shape[1]
    taps = sorted(a * pitch_out + b for a in (-1, 0) for b in (-1, 0))
    left = taps[0]
    dense = np.zeros((cout, 4 * cin, taps[-1] - left + 1), dtype=np.float32)
    for dt in (-1, 0, 1):
        for df in (-1, 0, 1):
            a, pt = (-1, 1) if dt < 0 else (0, dt)
            b, pf = (-1, 1) if df < 0 else (0, df)
            ph = pt * 2 + pf
            dense[:, ph * cin:(ph + 1) * cin, a * pitch_out + b - left] = w[:, :, df + 1, dt + 1]
    return taps, left, dense


def emit_conv_bn(x, conv, bn, relu):
    """conv (3x3 pad 1 or 1x1, stride 1|2, no bias) -> [eval BN] [-> ReLU] on a rank-4 grid Sym (bn=None: the bare
    convolution, as the second one of a pre-activation block)."""
    g = x.graph
    spec = g.grid_spec(x.view.tid)
    if spec is None or x.rank != 4:
        raise _ir.TraceError("2-D convolution applied to a tensor that is not a [B, C, F, T] grid")
    k, stride = conv.kernel_size[0], conv.stride[0]
    if conv.kernel_size not in ((3, 3), (1, 1)) or conv.stride[0] != conv.stride[1] or stride not in (1, 2) or conv.groups != 1 \
            or conv.dilation != (1, 1) or conv.padding != (k // 2, k // 2) or conv.bias is not None:
        raise _ir.TraceError("Conv2d%r: only 3x3/pad 1 and 1x1 kernels with stride 1 or 2 are implemented" % (conv,))
    if bn is not None:
        scale, shift = _folded_bn(bn)
    else:
        scale, shift = np.ones(conv.out_channels, dtype=np.float32), np.zeros(conv.out_channels, dtype=np.float32)
    w = _np(conv.weight).astype(np.float64) * scale.astype(np.float64)[:, None, None, None]       # [Cout, Cin, kF, kT]
    cout, cin = w.shape[0], w.shape[1]
    half = k // 2
    pos = [(dt, df) for df in range(-half, half + 1) for dt in range(-half, half + 1)]             # (time, frequency) offsets
    act = "relu" if relu else None
    if stride == 1:
        pitch = spec[3]
        taps = sorted(dt * pitch + df for dt, df in pos)
        left = taps[0]
        dense = np.zeros((cout, cin, taps[-1] - left + 1), dtype=np.float32)
        for dt, df in pos:
            dense[:, :, dt * pitch + df - left] = w[:, :, df + half, dt + half]
        inp = x.view
        if cin % _ir.CHAN_ALIGN != 0 and x.view.channels == cin:
            pass                                                    # e.g. the 1-channel head: the buffer pitch is zero padded
        out = g.tdnn(inp, dense, shift, taps, left, act1=act)
    elif cin in S2D_CIN:
        # Stride 2 as "space to depth": ONE gather brings the four phases (2t'+pt, 2f'+pf), pt, pf in {0, 1}, of the input
        # grid side by side into the output grid ([rows'][phase * cin + c]: 4 cin columns instead of the 9 cin of an im2col of
        # all taps, and both stride-2 convolutions of a down-sampling block - the 3 x 3 one and the 1 x 1 shortcut - read the
        # same tensor).  Input row 2t' + dt is phase pt = |dt| of output row t' + (dt < 0 ? -1 : 0), so the 3 x 3 kernel
        # becomes a 2 x 2 one on the output grid, taps (a, b) in {-1, 0}^2 = row offsets a * pitch' + b, with 9 of its
        # 16 (tap, phase) weight blocks filled; the 1 x 1 shortcut is a 1-tap layer on the phase (0, 0) channel slice.
        # Zero padding: rows above / left of the utterance are gap rows / the gap column of the output grid, input positions
        # beyond the utterance are zero-filled by the gather - the same zeros as the reference's padding=1.
        phases = [(0, 0), (0, 1), (1, 0), (1, 1)]                   # (pt, pf), phase index = pt * 2 + pf
        cache = g.__dict__.setdefault("_s2d_cache", {})
        key = x.view.key()
        if key not in cache:
            cache[key] = g.im2col(x.view, phases, stride)           # [rows'][phase * cin + c]
        cols = cache[key]
        if k == 1:
            out = g.tdnn(_ir.View(cols.tid, 0, cin), w[:, :, 0, 0].astype(np.float32)[:, :, None], shift, [0], 0, act1=act)
        else:
            taps, left, dense = s2d_kernel(w, g.grid_spec(cols.tid)[3])
            out = g.tdnn(cols, dense, shift, taps, left, act1=act)
            g.ops[-1].alg_fraction = 9.0 / 16.0                      # 9 of the 16 (tap, phase) blocks carry weights: FLOP accounting
    else:
        cols = g.im2col(x.view, pos, stride)                        # [rows'][k*cin + c]
        flat = np.zeros((cout, cin * len(pos), 1), dtype=np.float32)
        for i, (dt, df) in enumerate(pos):
            flat[:, i * cin:(i + 1) * cin, 0] = w[:, :, df + half, dt + half]
        out = g.tdnn(cols, flat, shift, [0], 0, act1=act)
    return _ir.Sym(g, out, 4)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, Conv=nn.Conv2d, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None,
                 norm_layer_params={}, full_pre_activation=True, use_se=False, se_ratio=4):
        super(BasicBlock, self).__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError("BasicBlock only supports groups=1 and base_width=64")
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.downsample = downsample
        self.stride = stride
        self.full_pre_activation = full_pre_activation
        if full_pre_activation:                          # resnet.py:59-68: BN-ReLU-conv, BN-ReLU-conv (module order = state_dict order)
            self.bn1 = norm_layer(inplanes, **norm_layer_params)
            self.relu1 = nn.ReLU(inplace=True)
            self.conv1 = conv3x3(inplanes, planes, Conv, stride)
            self.bn2 = norm_layer(planes, **norm_layer_params)
            self.relu2 = nn.ReLU(inplace=True)
            self.conv2 = conv3x3(planes, planes, Conv)
        else:                                            # resnet.py:46-56
            self.conv1 = conv3x3(inplanes, planes, Conv, stride)
            self.bn1 = norm_layer(planes, **norm_layer_params)
            self.relu1 = nn.ReLU(inplace=True)
            self.conv2 = conv3x3(planes, planes, Conv)
            self.bn2 = norm_layer(planes, **norm_layer_params)
            self.relu2 = nn.ReLU(inplace=True)
        self.se = SEBlock_2D(planes, se_ratio) if use_se else nn.Identity()

    def forward(self, x):
        if not isinstance(x, _ir.Sym):
            raise NotImplementedError("BasicBlock.forward() on a torch tensor: eager forward is not part of asv-subtools_amd")
        g = x.graph
        identity = x if self.downsample is None else emit_conv_bn(x, self.downsample[0], self.downsample[1], relu=False)
        if self.full_pre_activation:
            # resnet.py:87-104: relu(bn1(x)) is one elementwise pass (gap rows stay zero: the next convolution pads with
            # zeros AFTER the activation); bn2 + ReLU follow conv1 directly and fold into its weights / epilogue; conv2 is bare
            s1, t1 = _folded_bn(self.bn1)
            a = _ir.Sym(g, g.eltwise(x.view, scale=s1, shift=t1, act="relu"), 4)
            y = emit_conv_bn(a, self.conv1, self.bn2, relu=True)
            y = emit_conv_bn(y, self.conv2, None, relu=False)
        else:
            y = emit_conv_bn(x, self.conv1, self.bn1, relu=True)
            y = emit_conv_bn(y, self.conv2, self.bn2, relu=False)
        seg_scale = None
        if isinstance(self.se, SEBlock_2D):
            spec = g.grid_spec(y.view.tid)
            # mean over (F, T): the device pools over all rows of the segment incl. the zero row of each frame,
            # i.e. sum / (frames * pitch); the factor pitch / width is folded into fc_1
            m = g.pool(y.view, stddev=False)
            w1 = _np(self.se.fc_1.weight)[:, :, None] * np.float32(spec[3] / float(spec[2]))
            h = g.tdnn(m, w1, _np(self.se.fc_1.bias), [0], 0, act1="relu")
            seg_scale = g.tdnn(h, _np(self.se.fc_2.weight)[:, :, None], _np(self.se.fc_2.bias), [0], 0, act1="sigmoid")
        out = g.eltwise(y.view, b=identity.view, seg_scale=seg_scale, act=None if self.full_pre_activation else "relu")
        return _ir.Sym(g, out, 4)


def _emit_se_and_join(g, y, identity, se, pre_activation):
    """SE scale (resnet.py: SEBlock_2D) + residual (+ the final ReLU of the original form) as one elementwise pass."""
    seg_scale = None
    if isinstance(se, SEBlock_2D):
        spec = g.grid_spec(y.view.tid)
        # mean over (F, T): the device pools over all rows of the segment incl. the zero row of each frame,
        # i.e. sum / (frames * pitch); the factor pitch / width is folded into fc_1
        m = g.pool(y.view, stddev=False)
        w1 = _np(se.fc_1.weight)[:, :, None] * np.float32(spec[3] / float(spec[2]))
        h = g.tdnn(m, w1, _np(se.fc_1.bias), [0], 0, act1="relu")
        seg_scale = g.tdnn(h, _np(se.fc_2.weight)[:, :, None], _np(se.fc_2.bias), [0], 0, act1="sigmoid")
    return _ir.Sym(g, g.eltwise(y.view, b=identity.view, seg_scale=seg_scale, act=None if pre_activation else "relu"), 4)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (stride) -> 1x1 (x4) residual block in both forms of the reference (resnet.py:113-208), optional SE: the
    same fused pieces as BasicBlock - every convolution carries its eval BatchNorm (and ReLU) in weights / epilogue, the
    pre-activation form starts with one elementwise relu(bn1(x)) pass."""
    expansion = 4

    def __init__(self, inplanes, planes, Conv=nn.Conv2d, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None,
                 norm_layer_params={}, full_pre_activation=False, use_se=False, se_ratio=4):
        super(Bottleneck, self).__init__()
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        if groups != 1 or dilation > 1:
            raise NotImplementedError("grouped / dilated Bottleneck blocks are not implemented on the MI355X path")
        width = int(planes * (base_width / 64.)) * groups
        self.downsample = downsample
        self.stride = stride
        self.full_pre_activation = full_pre_activation
        if full_pre_activation:                          # resnet.py:149-162 (module order = state_dict order)
            self.bn1 = norm_layer(inplanes, **norm_layer_params)
            self.relu1 = nn.ReLU(inplace=True)
            self.conv1 = conv1x1(inplanes, width, Conv)
            self.bn2 = norm_layer(width, **norm_layer_params)
            self.relu2 = nn.ReLU(inplace=True)
            self.conv2 = conv3x3(width, width, Conv, stride)
            self.bn3 = norm_layer(width, **norm_layer_params)
            self.relu3 = nn.ReLU(inplace=True)
            self.conv3 = conv1x1(width, planes * self.expansion, Conv)
        else:                                            # resnet.py:133-147
            self.conv1 = conv1x1(inplanes, width, Conv)
            self.bn1 = norm_layer(width, **norm_layer_params)
            self.relu1 = nn.ReLU(inplace=True)
            self.conv2 = conv3x3(width, width, Conv, stride)
            self.bn2 = norm_layer(width, **norm_layer_params)
            self.relu2 = nn.ReLU(inplace=True)
            self.conv3 = conv1x1(width, planes * self.expansion, Conv)
            self.bn3 = norm_layer(planes * self.expansion, **norm_layer_params)
            self.relu3 = nn.ReLU(inplace=True)
        self.se = SEBlock_2D(planes * self.expansion, se_ratio) if use_se else nn.Identity()

    def forward(self, x):
        if not isinstance(x, _ir.Sym):
            raise NotImplementedError("Bottleneck.forward() on a torch tensor: eager forward is not part of asv-subtools_amd")
        g = x.graph
        identity = x if self.downsample is None else emit_conv_bn(x, self.downsample[0], self.downsample[1], relu=False)
        if self.full_pre_activation:                     # resnet.py:185-202
            s1, t1 = _folded_bn(self.bn1)
            a = _ir.Sym(g, g.eltwise(x.view, scale=s1, shift=t1, act="relu"), 4)
            y = emit_conv_bn(a, self.conv1, self.bn2, relu=True)
            y = emit_conv_bn(y, self.conv2, self.bn3, relu=True)
            y = emit_conv_bn(y, self.conv3, None, relu=False)
        else:                                            # resnet.py:164-183
            y = emit_conv_bn(x, self.conv1, self.bn1, relu=True)
            y = emit_conv_bn(y, self.conv2, self.bn2, relu=True)
            y = emit_conv_bn(y, self.conv3, self.bn3, relu=False)
        return _emit_se_and_join(g, y, identity, self.se, self.full_pre_activation)


class ResNet(nn.Module):
    """pre-conv + 4 residual stages, no avg-pool / fc (reference resnet.py:212-371)."""

    def __init__(self, head_inplanes, block="BasicBlock", layers=[3, 4, 6, 3], planes=[32, 64, 128, 256], convXd=2, full_pre_activation=True,
                 use_se=False, se_ratio=4, head_conv=True, head_conv_params={"kernel_size": 3, "stride": 1, "padding": 1}, head_maxpool=True,
                 head_maxpool_params={"kernel_size": 3, "stride": 1, "padding": 1}, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, norm_layer_params={}):
        super(ResNet, self).__init__()
        if convXd != 2:
            raise NotImplementedError("convXd=%r: only the 2-D trunk is implemented on the MI355X path" % (convXd,))
        if block not in ("BasicBlock", "Bottleneck"):
            raise NotImplementedError("block=%r: BasicBlock and Bottleneck are implemented on the MI355X path" % (block,))
        self.block = Bottleneck if block == "Bottleneck" else BasicBlock
        if head_maxpool or not head_conv or head_conv_params != {"kernel_size": 3, "stride": 1, "padding": 1}:
            raise NotImplementedError("only the ResNetXvector head (3x3 conv, stride 1, no max-pool) is implemented on the MI355X path")
        if replace_stride_with_dilation not in (None, [False, False, False]) or groups != 1:
            raise NotImplementedError("dilated / grouped ResNet variants are not implemented on the MI355X path")
        norm_layer = norm_layer or nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.norm_layer_params = norm_layer_params
        self.full_pre_activation = full_pre_activation
        self.head_conv, self.head_maxpool = head_conv, head_maxpool
        self.Conv = nn.Conv2d
        self.inplanes = planes[0]
        self.conv1 = nn.Conv2d(head_inplanes, self.inplanes, bias=False, **head_conv_params)
        self.bn1 = norm_layer(self.inplanes, **norm_layer_params)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = None
        self.downsample_multiple = head_conv_params["stride"]
        self.layer1 = self._make_layer(planes[0], layers[0], 1, use_se, se_ratio)
        self.layer2 = self._make_layer(planes[1], layers[1], 2, use_se, se_ratio)
        self.layer3 = self._make_layer(planes[2], layers[2], 2, use_se, se_ratio)
        self.layer4 = self._make_layer(planes[3], layers[3], 2, use_se, se_ratio)
        self.downsample_multiple *= 8
        self.output_planes = planes[3] * self.block.expansion

    def _make_layer(self, planes, blocks, stride, use_se, se_ratio):
        downsample = None
        block = self.block
        if stride != 1 or self.inplanes != planes * block.expansion:       # resnet.py:318-322
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, nn.Conv2d, stride),
                                       self._norm_layer(planes * block.expansion, **self.norm_layer_params))
        kw = dict(norm_layer=self._norm_layer, norm_layer_params=self.norm_layer_params, full_pre_activation=self.full_pre_activation,
                  use_se=use_se, se_ratio=se_ratio)
        seq = [block(self.inplanes, planes, nn.Conv2d, stride, downsample, **kw)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes, nn.Conv2d, **kw) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def get_downsample_multiple(self):
        return self.downsample_multiple

    def get_output_planes(self):
        return self.output_planes

    def forward(self, x):
        if not isinstance(x, _ir.Sym):
            raise NotImplementedError("ResNet.forward() on a torch tensor: eager forward is not part of asv-subtools_amd")
        x = emit_conv_bn(x, self.conv1, self.bn1, relu=True)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
        return x
