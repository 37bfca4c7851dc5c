# -*- coding:utf-8 -*-
"""`TopVirtualNnet` and the `for_extract_embedding` decorator - the drop-in boundary
(reference libs/nnet/framework.py:12-55 and 61-186; contract in SURVEY.md section 8(b)).

Same call: `model.extract_embedding(feats[T, D]) -> 1-D CPU float32 tensor`, same chunking
rule (ceil(T/maxChunk) near-equal chunks, frame-weighted mean).  Different execution: the
decorated body is recorded once into a layer program and run by libasv_amd.so on the
MI355X; `extract_embedding_batch` exposes the batched path the per-utterance reference
loop (pipeline/onestep/extract_embeddings.py:73-83) does not have.
"""

import numpy as np
import torch

import libs.support.utils as utils


def _to_frames_matrix(input, isMatrix):
    """Reference input conventions -> float32 [T, D] numpy (framework.py:28-31)."""
    if isinstance(input, torch.Tensor):
        input = input.detach().cpu().numpy()
    x = np.asarray(input, dtype=np.float32)
    if isMatrix:
        if x.ndim != 2:
            raise ValueError("extract_embedding expects a [frames, feature-dim] matrix, got shape %s" % (x.shape,))
        return x
    if x.ndim != 3 or x.shape[0] != 1:
        raise ValueError("extract_embedding(isMatrix=False) expects a [1, feature-dim, frames] tensor, got %s" % (x.shape,))
    return np.ascontiguousarray(x[0].T)


def for_extract_embedding(maxChunk=10000, isMatrix=True):
    """Decorator for a model's `extract_embedding(self, inputs)` body."""
    def wrapper(function):
        def _wrapper(self, input):
            x = _to_frames_matrix(input, isMatrix)
            emb = self._amd_engine(function).extract_batch([x], max_chunk=maxChunk)
            return emb[0]                       # 1-D CPU float32 tensor, like framework.py:52
        _wrapper.__wrapped_body__ = function
        _wrapper.max_chunk = maxChunk
        _wrapper.is_matrix = isMatrix
        _wrapper.__doc__ = function.__doc__
        return _wrapper
    return wrapper


class TopVirtualNnet(torch.nn.Module):
    """Base class of every model blueprint: implement `init()` (not `__init__`) and
    `extract_embedding()`; `forward/get_loss` are training-side and not provided here."""

    def __init__(self, *args, **kwargs):
        super(TopVirtualNnet, self).__init__()
        self.model_creation = "{0}({1},{2})".format(type(self).__name__, utils.iterator_to_params_str(args),
                                                    utils.dict_to_params_str(kwargs))
        self.loss = None
        self.use_step = False
        self.transform_keys = []
        self.rename_transform_keys = {}
        self._amd_engines = {}
        self.init(*args, **kwargs)

    def init(self, *args, **kwargs):
        raise NotImplementedError

    def get_model_creation(self):
        return self.model_creation

    def forward(self, *inputs):
        raise NotImplementedError("training-time forward() is outside asv-subtools_amd; use extract_embedding()")

    def get_loss(self, *inputs, targets=None):
        raise NotImplementedError("training is outside asv-subtools_amd")

    def auto(self, layer, x):
        """`layer(x) if layer is not None else x`"""
        return layer(x) if layer is not None else x

    def load_transform_state_dict(self, state_dict):
        assert isinstance(self.transform_keys, list)
        assert isinstance(self.rename_transform_keys, dict)
        remaining = {utils.key_to_value(self.rename_transform_keys, k, False): v for k, v in state_dict.items()
                     if k.split(".")[0] in self.transform_keys or k in self.transform_keys}
        self.load_state_dict(remaining, strict=False)
        return self

    # ---- compiled-engine cache ----------------------------------------------------------
    def _invalidate_engines(self):
        for eng in getattr(self, "_amd_engines", {}).values():
            eng.close()
        self.__dict__["_amd_engines"] = {}

    def load_state_dict(self, *args, **kwargs):
        self._invalidate_engines()              # device weights are a snapshot of the parameters
        return super(TopVirtualNnet, self).load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._invalidate_engines()              # .cuda() / .cpu() / .to(): recompile for the new device
        return super(TopVirtualNnet, self)._apply(fn, *args, **kwargs)

    def _amd_engine(self, function=None, replica=0):
        """Compiles (once per device / precision / body / replica) and returns the libasv_amd engine.  `replica` > 0: further engines
        of the same program (own activation arena, own HIP stream in the caller's hands): what the extraction script and bench.py
        rotate consecutive batches over; cached here like the first one, dropped with it by load_state_dict() / .to()."""
        from libs.amd import engine as _engine
        if function is None:
            function = getattr(type(self).extract_embedding, "__wrapped_body__", None)
        precision = getattr(self, "amd_precision", None) or _engine.default_precision()
        p = next(self.parameters())
        key = (function, str(p.device), precision, _engine.default_flags(), _engine.gather_fuse_on(), getattr(self, "extracted_embedding", None), int(replica))
        eng = self._amd_engines.get(key)
        if eng is None:
            eng = _engine.compile_model(self, function=function, precision=precision)
            self._amd_engines[key] = eng
        return eng

    @for_extract_embedding(maxChunk=10000, isMatrix=True)
    def extract_embedding(self, inputs):
        raise NotImplementedError

    def extract_embedding_batch(self, mats, max_chunk=None):
        """Batched superset of extract_embedding: list of [T_i, D] matrices -> CPU tensor [B, E]."""
        method = type(self).extract_embedding
        if max_chunk is None:
            max_chunk = getattr(method, "max_chunk", 10000)
        return self._amd_engine().extract_batch(list(mats), max_chunk=max_chunk)

    def embedding_dim(self):
        return self._amd_engine().embed_dim

    def predict(self, outputs):
        with torch.no_grad():
            return torch.squeeze(torch.argmax(outputs, dim=1))

    def step(self, epoch, this_iter, epoch_batchs):
        pass

    def backward_step(self, epoch, this_iter, epoch_batchs):
        pass
