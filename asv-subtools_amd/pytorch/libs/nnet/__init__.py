# -*- coding:utf-8 -*-
"""`from libs.nnet import *` - the import every reference model blueprint performs
(/root/reference/pytorch/libs/nnet/__init__.py:5-13, model/xvector.py:12).

Exports the extraction-path classes under the reference's names; everything training-side
resolves to a placeholder that raises when constructed (libs/nnet/loss.py)."""

import importlib as _importlib

# order matters only for readability: boundary first, then layers, pooling, 2-D trunk, placeholders
_SUBMODULES = ("framework", "activation", "components", "dropout", "pooling", "resnet", "loss")

for _name in _SUBMODULES:
    _mod = _importlib.import_module("." + _name, __name__)
    globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("_")})

del _importlib, _name, _mod
