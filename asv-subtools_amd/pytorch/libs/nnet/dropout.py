# -*- coding:utf-8 -*-
"""Dropout / augmentation names some blueprints construct (reference libs/nnet/dropout.py:239-290).  Extraction runs in
eval mode where every one of them is the identity, so the wrapper hands out plain torch modules (or None for p = 0, as
the reference does) - they are never called on the symbolic trace because `TopVirtualNnet.auto` skips None and the
blueprints' `extract_embedding` bodies do not route through them."""

import torch


def get_dropout_from_wrapper(p=0., dropout_params={}):
    if not 0. <= p < 1.:
        raise ValueError("dropout probability %r outside [0, 1)" % (p,))
    if p == 0:
        return None
    dim = int(dict(dropout_params).get("dim", 2))
    inplace = bool(dict(dropout_params).get("inplace", True))
    return {1: torch.nn.Dropout, 2: torch.nn.Dropout2d, 3: torch.nn.Dropout3d}.get(dim, torch.nn.Dropout)(p=p, inplace=inplace)
