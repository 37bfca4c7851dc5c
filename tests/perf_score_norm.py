#!/usr/bin/env python3
"""(lives under tests/ because it times the oracle as well: only tests/ may import oracle/)
Throughput of the device score normalisation (asv_score_norm) at VoxCeleb1-O scale, next to the numpy oracle on the
host (the reference itself is a pandas groupby + a Python loop over the trials).  Prints one JSON line.

    python tests/perf_score_norm.py [--enroll 4708] [--test 4708] [--cohort 3000] [--trials 37720] [--top-n 300]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "asv-subtools_amd", "pytorch"))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--enroll", type=int, default=4708)
    ap.add_argument("--test", type=int, default=4708)
    ap.add_argument("--cohort", type=int, default=3000)
    ap.add_argument("--trials", type=int, default=37720)
    ap.add_argument("--top-n", type=int, default=300)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import torch
    from libs.amd import scoring
    from oracle import scoring_oracle as S
    rng = np.random.RandomState(0)
    e = rng.standard_normal((args.enroll, args.dim)).astype(np.float32)
    t = rng.standard_normal((args.test, args.dim)).astype(np.float32)
    c = rng.standard_normal((args.cohort, args.dim)).astype(np.float32)
    ei = rng.randint(0, args.enroll, args.trials).astype(np.int32)
    ti = rng.randint(0, args.test, args.trials).astype(np.int32)
    dev = torch.device("cuda", 0)
    ed, td, cd = (scoring.length_normalize(torch.from_numpy(x).to(dev)) for x in (e, t, c))
    eid, tid = torch.from_numpy(ei).to(dev), torch.from_numpy(ti).to(dev)
    res = {}
    for cross in (False, True):
        def run():
            raw = scoring.score_trials(ed, td, eid, tid)
            return scoring.score_normalize(raw, scoring.score_matrix(ed, cd), scoring.score_matrix(td, cd), eid, tid, top_n=args.top_n, cross_select=cross)
        out = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            out = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        en, tn, cn = ed.cpu().numpy(), td.cpu().numpy(), cd.cpu().numpy()
        t1 = time.perf_counter()
        raw = np.einsum("ij,ij->i", en[ei], tn[ti]).astype(np.float32)
        want = S.score_norm((en @ cn.T).astype(np.float32), (tn @ cn.T).astype(np.float32), ei, ti, raw, args.top_n, cross)
        dt_cpu = time.perf_counter() - t1
        err = float(np.abs(out.cpu().numpy() - want).max())
        res["cross_select" if cross else "plain"] = {"device_ms": round(dt * 1e3, 3), "trials_per_s": round(args.trials / dt), "numpy_oracle_s": round(dt_cpu, 3),
                                                     "max_abs_diff_vs_oracle": err}
    print(json.dumps({"workload": "cosine (trials + enrol x cohort + test x cohort) + AS-norm top-%d, %d enrol, %d test, %d cohort, %d trials, dim %d"
                      % (args.top_n, args.enroll, args.test, args.cohort, args.trials, args.dim), **res}))


if __name__ == "__main__":
    main()
