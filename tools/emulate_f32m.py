#!/usr/bin/env python3
"""CPU emulation of the precision forms of the f32x mode on the standard x-vector (no device): how far are the embeddings from the f32
forward when the matrix products of the frame layers run as

    f16        one half product (both operands rounded to IEEE half)
    f32x       w_hi x_hi + w_hi x_lo + w_lo x_hi, all halves                      (three 16-bit matrix instructions)
    noxlo / nowlo   two of the three                                               (the measured "two are not enough" variants)
    f32m       w_hi x_hi exact + [e4m3(w_hi) e5m2(x_lo)] + [e4m3(w_lo) e5m2(x_hi)]  (one 16-bit + half of a block-scaled 8-bit instruction:
               asv-subtools_amd/csrc/kernels_tdnn_chainm.hip, kernels_tdnn_x3m.hip)
    f32m-52    the same with e5m2 weights

Every product is evaluated in float64 on the rounded operands (the f32 accumulation of the device adds ~1e-7), activations are stored as
f32 between the layers, the pooled layers stay exact - the numpy oracle (oracle/np_oracle.py) with its matrix product replaced.  Calibration
against the device (profiles/r5y_pytest.txt): f16 2.8e-4, noxlo 9.6e-5, nowlo 2.3e-4 measured there; this script gives 2.5 - 3.4e-4,
8.3 - 8.7e-5, 2.2 - 3.3e-4 for its three weight seeds.  The north star's gate is 1e-4.

    python tools/emulate_f32m.py [--seeds 0,1,2] [--utts 16]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "asv-subtools_amd", "pytorch"), os.path.join(REPO, "tests")]
import helpers                                     # noqa: E402
from libs.amd import synth                         # noqa: E402
from oracle import np_oracle as O                  # noqa: E402  (this tool IS a checker: it never runs in the product path)


def f16(a):
    return a.astype(np.float16).astype(np.float64)


def e4m3(a):
    return torch.from_numpy(np.clip(a, -448, 448).astype(np.float32)).to(torch.float8_e4m3fn).to(torch.float32).numpy().astype(np.float64)


def e5m2(a):
    return torch.from_numpy(np.clip(a, -57344, 57344).astype(np.float32)).to(torch.float8_e5m2).to(torch.float32).numpy().astype(np.float64)


MODE = "f32"


def product(x, w):
    """x [T, K], w [K, N]: the mode's matrix product."""
    x = x.astype(np.float32).astype(np.float64)
    w = w.astype(np.float64)
    if MODE == "f32":
        return x @ w
    if MODE == "f16":
        return f16(x) @ f16(w)
    s = 2.0 ** (14 - np.frexp(np.abs(w).max())[1])          # runtime.hip x3_weight_scale: the largest weight in [2^13, 2^14)
    ws = w * s
    wh = f16(ws)
    wl = ws - wh
    xh = f16(x)
    xl = x - xh
    if MODE == "f32x":
        return (xh @ wh + xl @ wh + xh @ f16(wl)) / s
    if MODE == "noxlo":
        return (xh @ wh + xh @ f16(wl)) / s
    if MODE == "nowlo":
        return (xh @ wh + xl @ wh) / s
    if MODE in ("f32m", "f32m-52"):
        wq = e4m3 if MODE == "f32m" else e5m2
        a, b = (-6, 6) if MODE == "f32m" else (0, 11)
        return (xh @ wh + (e5m2(xl * 2.0 ** 11) @ wq(wh * 2.0 ** a)) * 2.0 ** (-a - 11) + (e5m2(x) @ wq(wl * 2.0 ** b)) * 2.0 ** (-b)) / s
    raise ValueError(MODE)


def tdnn_affine(x, weight, bias, context, pad=True):
    left = context[0] if context[0] < 0 else 0
    right = context[-1] if context[-1] > 0 else 0
    T = x.shape[0]
    xp = np.zeros((T - left + right, x.shape[1]), dtype=x.dtype)
    xp[-left:-left + T] = x
    xs = np.concatenate([xp[off - left:off - left + T] for off in context], axis=1)
    w = np.concatenate([weight[:, :, off - left].T for off in context], axis=0)
    y = xs.astype(np.float64) @ w.astype(np.float64) if (T == 1 or MODE == "f32") else product(xs, w)      # pooled layers: the exact f32 path
    if bias is not None:
        y = y + bias.astype(np.float64)
    return y.astype(np.float32).astype(np.float64)


def main():
    global MODE
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0,1,2")
    ap.add_argument("--utts", type=int, default=16)
    args = ap.parse_args()
    O.tdnn_affine = tdnn_affine
    model = helpers.build_model("xvector.py", "Xvector(80,10,training=False)")
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    mats = [synth.synth_feats(200, 80, i) for i in range(args.utts)]
    for seed in (int(s) for s in args.seeds.split(",")):
        sd64 = O.cast_state_dict(synth.synth_state_dict(shapes, seed), np.float64)
        t0, res = time.time(), {}
        for MODE in ("f32", "f16", "f32x", "noxlo", "nowlo", "f32m", "f32m-52"):
            res[MODE] = np.stack([O.xvector_embed(m.astype(np.float64), sd64, "far") for m in mats])
        ref = res.pop("f32")
        print("weight seed %d: max |e - e_f32| / max |e_f32| over %d utterances:  " % (seed, len(mats)) +
              "  ".join("%s %.3g" % (k, np.abs(v - ref).max() / np.abs(ref).max()) for k, v in res.items()) + "   (%.0f s)" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
