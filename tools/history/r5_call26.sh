#!/bin/bash
# ragged table (U[200, 1000] frames) through the script: stream vs --sharded, rates + the two arks compared
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 200 python tools/bench_pipeline.py --utts 20000 --lengths 200,1000 --precisions f32x,bf16 --paths stream,sharded --once > $out/r5zd_ark_ragged.json 2>$out/r5zd_ark_ragged.err
python - <<PY
import json
d=json.load(open("$out/r5zd_ark_ragged.json"))
print(d["workload"])
for v in d["all_runs"]: print(v["run"], v.get("loop_utts_per_s"), v.get("complete"), v.get("consumer_thread_seconds"), v.get("error","")[:300])
PY
timeout 150 python - <<'PY' 2>&1 | tail -6
import os, sys, subprocess, numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, os.path.join(root, "tools")); sys.path.insert(0, os.path.join(root, "asv-subtools_amd", "pytorch"))
import bench_pipeline as bp
from libs.support import kaldi_io
d = "/tmp/asv_pipe_cmp"
f = bp.prepare(d, 3000, lengths=(200, 1000))
outs = {}
for path in ("stream", "sharded"):
    o = os.path.join(d, "x_%s.ark" % path)
    rspec = "ark:" + f["ark"] if path == "stream" else "scp:" + f["scp"]
    extra = ["--sharded", "true"] if path == "sharded" else []
    r = subprocess.run([sys.executable, bp.SCRIPT, "--nnet-config", f["cfg"], "--use-gpu", "true", "--gpu-id", "0"] + extra + [f["params"], rspec, "ark:" + o],
                       capture_output=True, text=True, env=dict(os.environ, ASV_AMD_PRECISION="f32x", ASV_AMD_SHARD_SEGMENT="700"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-800:]
    outs[path] = list(kaldi_io.read_vec_flt_ark(o))
a, b = outs["stream"], outs["sharded"]
assert [k for k, _ in a] == [k for k, _ in b] and len(a) == 3000
A, B = np.stack([v for _, v in a]), np.stack([v for _, v in b])
rel = np.abs(A - B).max(axis=1) / np.abs(A).max(axis=1)
print("ragged table, 3000 utterances, f32x: stream vs --sharded (segments of 700): same keys in the same order; max relative difference per vector: max %.3g, median %.3g; bit-identical vectors: %d" % (rel.max(), np.median(rel), int((A == B).all(axis=1).sum())))
PY
