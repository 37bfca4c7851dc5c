# -*- coding:utf-8 -*-
"""Activation factory (reference libs/nnet/activation.py:81-106).  Only the activations that
the HIP epilogue implements are constructible on the extraction path."""

import torch

HIP_ACTIVATIONS = ("relu", "tanh", "sigmoid")


def Nonlinearity(nonlinearity="relu", inplace=True, negative_slope=0.01):
    """Returns a torch module used as a *marker* (its class tells the recorder which epilogue
    activation to fuse) or None for '' / None / False."""
    if nonlinearity == "relu":
        return torch.nn.ReLU(inplace=inplace)
    if nonlinearity == "tanh":
        return torch.nn.Tanh()
    if nonlinearity == "sigmoid":
        return torch.nn.Sigmoid()
    if nonlinearity == "" or nonlinearity is None or nonlinearity is False:
        return None
    if nonlinearity in ("leaky_relu", "selu", "mish", "swish", "gelu", "double_swish"):
        raise NotImplementedError("nonlinearity '{0}' exists in the reference but is not implemented by the "
                                  "MI355X extraction kernels (have: {1})".format(nonlinearity, ", ".join(HIP_ACTIVATIONS)))
    raise ValueError("Do not support {0} nonlinearity now.".format(nonlinearity))


def activation_name(module):
    """Marker module -> epilogue activation name."""
    if module is None:
        return None
    for cls, name in ((torch.nn.ReLU, "relu"), (torch.nn.Tanh, "tanh"), (torch.nn.Sigmoid, "sigmoid")):
        if isinstance(module, cls):
            return name
    raise NotImplementedError("activation module %r has no HIP epilogue" % (module,))
