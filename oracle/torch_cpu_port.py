"""torch-CPU port of the reference's extraction path (TEST / BASELINE INFRASTRUCTURE - see
oracle/__init__.py).

This is what `cpu_baseline` in bench.py times on the GPU box's host cores: the same torch
calls the reference issues per utterance, batch = 1 - F.pad + (weight * mask) + F.conv1d with
the DENSE 5/7-tap kernel (libs/nnet/components.py:107-149), in-place ReLU, eval batch_norm
(410-431), mean / two-pass std pooling (libs/nnet/pooling.py:58-67), wrapped by the chunking
loop of libs/nnet/framework.py:18-52.  The reference tree itself cannot travel to the GPU
box, hence "port"; tests/test_oracle_golden.py pins it to the reference's own outputs.
"""

import numpy as np
import torch
import torch.nn.functional as F


class _Tdnn(object):
    def __init__(self, sd, prefix, context):
        self.w = torch.from_numpy(sd[prefix + ".affine.weight"])
        self.b = torch.from_numpy(sd[prefix + ".affine.bias"]) if prefix + ".affine.bias" in sd else None
        self.left = context[0] if context[0] < 0 else 0
        self.right = context[-1] if context[-1] > 0 else 0
        tot = self.right - self.left + 1
        self.mask = None
        if len(context) != tot:
            self.mask = torch.tensor([[[1 if i in context else 0 for i in range(self.left, self.right + 1)]]])
        bn = prefix + ".batchnorm"
        self.bn = None
        if bn + ".running_mean" in sd:
            g = torch.from_numpy(sd[bn + ".weight"]) if bn + ".weight" in sd else None
            b = torch.from_numpy(sd[bn + ".bias"]) if bn + ".bias" in sd else None
            self.bn = (torch.from_numpy(sd[bn + ".running_mean"]), torch.from_numpy(sd[bn + ".running_var"]), g, b)

    def affine(self, x):
        x = F.pad(x, (-self.left, self.right), mode="constant", value=0.0)
        filters = self.w * self.mask if self.mask is not None else self.w      # redone every call, like the reference
        return F.conv1d(x, filters, self.b, stride=1, padding=0, dilation=1, groups=1)

    def __call__(self, x, relu=True):
        x = self.affine(x)
        if relu:
            x = F.relu(x, inplace=True)
        if self.bn is not None:
            rm, rv, g, b = self.bn
            x = F.batch_norm(x, rm, rv, g, b, False, 0.1, 1e-5)
        return x


class XvectorCpu(object):
    """model/xvector.py:77-98 on CPU tensors."""
    CONTEXTS = {"tdnn1": [-2, -1, 0, 1, 2], "tdnn2": [-2, 0, 2], "tdnn3": [-3, 0, 3], "tdnn4": [0], "tdnn5": [0],
                "tdnn6": [0], "tdnn7": [0]}

    def __init__(self, sd, position="far"):
        self.layers = {n: _Tdnn(sd, n, c) for n, c in self.CONTEXTS.items()}
        self.position = position

    def _embed(self, x):
        for n in ("tdnn1", "tdnn2", "tdnn3", "tdnn4", "tdnn5"):
            x = self.layers[n](x)
        mean = x.mean(dim=2, keepdim=True)
        var = torch.sum((x - mean) ** 2, dim=2, keepdim=True) / x.shape[2]
        x = torch.cat((mean, torch.sqrt(var.clamp(min=1.0e-10))), dim=1)
        if self.position == "far":
            return self.layers["tdnn6"].affine(x)
        return self.layers["tdnn7"].affine(self.layers["tdnn6"](x))

    def extract_embedding(self, feats, max_chunk=10000):
        with torch.no_grad():
            x = torch.tensor(feats).unsqueeze(0).transpose(1, 2)
            T = x.shape[2]
            num_split = (T + max_chunk - 1) // max_chunk
            split = T // num_split
            off, stats = 0, 0.0
            for _ in range(num_split - 1):
                stats = stats + split * self._embed(x[:, :, off:off + split])
                off += split
            emb = (stats + (T - off) * self._embed(x[:, :, off:])) / T
            return torch.squeeze(emb.transpose(1, 2)).cpu()


def time_cpu_baseline(extractor, mats, budget_s=20.0, min_utts=8):
    """Times `extractor.extract_embedding` batch=1 over `mats` (cycled) for about `budget_s`
    seconds after one warm-up utterance.  Returns (utts_per_s, n_utts, seconds)."""
    import time
    extractor.extract_embedding(mats[0])
    n, t0 = 0, time.perf_counter()
    while True:
        extractor.extract_embedding(mats[n % len(mats)])
        n += 1
        dt = time.perf_counter() - t0
        if (dt >= budget_s and n >= min_utts) or n >= 100000:
            return n / dt, n, dt
