#!/usr/bin/env python3
"""ark -> ark throughput of the drop-in extraction script (host I/O + PCIe + device), next to the device-resident
number of bench.py.  Writes a synthetic Kaldi feature archive (N utterances x 200 x 80 f32), a checkpoint and an
nnet.config for the standard x-vector, runs asv-subtools_amd/pytorch/pipeline/onestep/extract_embeddings.py on it
with the reference's command line and prints one JSON line.

    python tools/bench_pipeline.py [--utts 20000] [--precision bf16] [--dir /tmp/asv_pipe]
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "asv-subtools_amd", "pytorch"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=20000)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--dir", default="/tmp/asv_pipe")
    args = ap.parse_args()
    import torch
    from libs.support import kaldi_io
    import libs.support.utils as utils
    from libs.amd import synth
    os.makedirs(args.dir, exist_ok=True)
    blueprint = os.path.join(REPO, "asv-subtools_amd", "pytorch", "model", "xvector.py")
    creation = "Xvector(80,10,training=False)"
    model = utils.create_model_from_py(blueprint, creation)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0)
    params = os.path.join(args.dir, "final.params")
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, params)
    cfg = os.path.join(args.dir, "nnet.config")
    utils.write_nnet_config(blueprint, creation, cfg)
    feats = os.path.join(args.dir, "feats.ark")
    base = [synth.synth_feats(args.frames, 80, 50_000 + i) for i in range(64)]
    t0 = time.perf_counter()
    with open(feats, "wb") as f:
        for i in range(args.utts):
            kaldi_io.write_mat(f, base[i % 64], key="utt%07d" % i)
    t_write = time.perf_counter() - t0
    out = os.path.join(args.dir, "xvector.ark")
    script = os.path.join(REPO, "asv-subtools_amd", "pytorch", "pipeline", "onestep", "extract_embeddings.py")
    env = dict(os.environ, ASV_AMD_PRECISION=args.precision)
    t0 = time.perf_counter()
    res = subprocess.run([sys.executable, script, "--nnet-config", cfg, "--use-gpu", "true", "--gpu-id", "0", params, "ark:" + feats, "ark:" + out],
                         capture_output=True, text=True, env=env)
    dt = time.perf_counter() - t0
    if res.returncode != 0:
        sys.exit(res.stdout + res.stderr)
    n = sum(1 for _ in kaldi_io.read_vec_flt_ark(out))
    assert n == args.utts, (n, args.utts)
    print(json.dumps({"workload": "%d utterances x %d x 80 f32 Kaldi ark -> x-vector ark, %s" % (args.utts, args.frames, args.precision),
                      "utts_per_s_end_to_end_incl_process_start": round(args.utts / dt, 1), "seconds": round(dt, 2),
                      "feature_ark_gb": round(os.path.getsize(feats) / 1e9, 3), "ark_write_seconds_synthetic": round(t_write, 2)}))
    print(res.stderr[-1500:], file=sys.stderr)


if __name__ == "__main__":
    main()
