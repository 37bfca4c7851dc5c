# -*- coding:utf-8 -*-
"""Training losses exist in the reference (libs/nnet/loss.py) and are referenced by name in
blueprint `init()` bodies behind `if training:`.  Extraction constructs models with
training=False, so these names only need to resolve; constructing one is a clear error."""


def _training_only(name):
    class _TrainingOnly(object):
        def __init__(self, *args, **kwargs):
            raise NotImplementedError("%s is a training-time component; asv-subtools_amd implements the "
                                      "embedding-extraction path only - build the model with training=False" % name)
    _TrainingOnly.__name__ = name
    return _TrainingOnly


SoftmaxLoss = _training_only("SoftmaxLoss")
MarginSoftmaxLoss = _training_only("MarginSoftmaxLoss")
MarginSoftmaxLoss_v1 = _training_only("MarginSoftmaxLoss_v1")
MixupLoss = _training_only("MixupLoss")
MarginWarm = _training_only("MarginWarm")
Mixup = _training_only("Mixup")
SpecAugment = _training_only("SpecAugment")
