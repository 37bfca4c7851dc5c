#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 400 python tools/bench_pipeline.py --utts 50000 --precisions f32x,bf16 --paths stream,sharded > $out/r5zb_ark.json 2>$out/r5zb_ark.err
python - <<PY
import json
d=json.load(open("$out/r5zb_ark.json"))
for v in d["all_runs"]: print(v["run"], v.get("loop_utts_per_s"), v.get("loop_seconds"), v.get("consumer_thread_seconds"), v.get("error","")[:300])
print({k: v.get("loop_utts_per_s") for k, v in d["runs"].items()})
PY
