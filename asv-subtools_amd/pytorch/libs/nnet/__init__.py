# -*- coding:utf-8 -*-
"""`from libs.nnet import *` - the import every reference model blueprint performs
(/root/reference/pytorch/libs/nnet/__init__.py:5-13, model/xvector.py:12).  Exports the
extraction-path classes under the reference's names."""

from .framework import *
from .activation import *
from .components import *
from .pooling import *
from .resnet import *
from .loss import *
