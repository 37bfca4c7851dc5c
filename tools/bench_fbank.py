#!/usr/bin/env python3
"""Front-end throughput: asv_fbank (+ asv_cmvn) over a batch of synthetic waveforms resident in HBM.
    python tools/bench_fbank.py [--utts 512] [--seconds 2.0] [--bins 80] [--iters 50]
Prints one JSON line: frames/s, x real time, algorithmic HBM GB/s (4 B/sample in + 4*dim B/frame out)."""
import argparse
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "asv-subtools_amd", "pytorch")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=512)
    ap.add_argument("--seconds", type=float, default=2.015)          # 200 frames: the C2 utterance
    ap.add_argument("--bins", type=int, default=80)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cmn", type=int, default=1)
    a = ap.parse_args()
    import torch
    from libs.amd import frontend, synth
    n = int(a.seconds * 16000)
    base = [torch.from_numpy(synth.synth_wave(n, 900 + i)).cuda() for i in range(8)]
    waves = [base[i % 8] for i in range(a.utts)]
    wave = torch.cat(waves)
    soff = np.arange(a.utts + 1, dtype=np.int64) * n
    feats, off = frontend.fbank_device(wave, soff, num_mel_bins=a.bins, mean_norm=bool(a.cmn))
    frames = int(off[-1])
    for _ in range(20):
        frontend.fbank_device(wave, soff, num_mel_bins=a.bins, mean_norm=bool(a.cmn))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        frontend.fbank_device(wave, soff, num_mel_bins=a.bins, mean_norm=bool(a.cmn))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    alg = 4.0 * n * a.utts + 4.0 * a.bins * frames * (3 if a.cmn else 1)
    print(json.dumps({"metric": "fbank_frames_per_sec", "value": frames / ms * 1e3, "ms_per_batch": ms, "utts": a.utts, "frames": frames,
                      "x_real_time": a.utts * a.seconds / (ms * 1e-3), "algorithmic_GBps": alg / ms * 1e-6, "bins": a.bins, "cmn": bool(a.cmn),
                      "note": "waveforms packed in HBM; includes the offsets upload of every call"}))


if __name__ == "__main__":
    main()
