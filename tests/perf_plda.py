#!/usr/bin/env python3
"""(lives under tests/ because it times the oracle as well: only tests/ may import oracle/)
PLDA training throughput: asv_plda_train (f64 statistics + EM on the device) on a planted-speaker set.
    python tests/perf_plda.py [--speakers 6000] [--per-speaker 30] [--dim 256] [--iters 10]
Prints one JSON line; --cpu-classes N also times the numpy oracle EM (one inverse per class, like the reference) on the
first N classes for scale."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "asv-subtools_amd", "pytorch"), REPO]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--speakers", type=int, default=6000)
    ap.add_argument("--per-speaker", type=int, default=30)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cpu-classes", type=int, default=0)
    a = ap.parse_args()
    import torch
    from libs.amd import scoring, synth
    x, labels = synth.synth_speaker_embeddings(a.speakers, a.per_speaker, a.dim, seed=5, within=1.0, between=0.6)
    sizes = np.random.RandomState(6).randint(max(2, a.per_speaker // 3), a.per_speaker + 1, size=a.speakers)
    keep = np.zeros(len(labels), dtype=bool)
    start = np.arange(a.speakers) * a.per_speaker
    for spk, n in enumerate(sizes):
        keep[start[spk]:start[spk] + n] = True
    x, labels = x[keep], labels[keep]
    xd = torch.from_numpy(x).cuda()
    scoring.train_plda(xd, labels, num_iters=1)                        # warm up (module load, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mean, within, between = scoring.train_plda(xd, labels, num_iters=a.iters)
    dt = time.perf_counter() - t0
    flops_stats = 2.0 * len(labels) * a.dim * a.dim
    res = {"metric": "plda_train_seconds", "value": dt, "vectors": int(len(labels)), "classes": a.speakers, "distinct_class_sizes": int(len(set(sizes.tolist()))),
           "dim": a.dim, "em_iters": a.iters, "dtype": "f64", "scatter_gflop": flops_stats * 1e-9,
           "note": "includes the host grouping of row indices by class and the D2H copy of mean / within / between"}
    if a.cpu_classes:
        from oracle import scoring_oracle as S
        st = S.PldaStats(a.dim)
        order = np.argsort(np.bincount(labels), kind="stable")[:a.cpu_classes]
        for spk in sorted(order, key=lambda k: (labels == k).sum()):
            st.add_samples(1.0, x[labels == spk].astype(np.float64))
        t0 = time.perf_counter()
        S.plda_em(st, 1)
        res["cpu_oracle_seconds_per_iter_per_class"] = (time.perf_counter() - t0) / a.cpu_classes
    print(json.dumps(res))


if __name__ == "__main__":
    main()
