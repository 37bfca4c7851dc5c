# -*- coding:utf-8 -*-
"""2-D ResNet trunk of ResNetXvector (reference libs/nnet/resnet.py:212-371).

Status: SURVEY.md section 8 rows a12-a14 (config C5) are not built yet; constructing the
trunk says so instead of silently running torch convolutions."""

import torch


class ResNet(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the ResNet34-SE 2-D trunk (config C5) is not implemented on the MI355X path yet")
