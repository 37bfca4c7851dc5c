#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 300 python tools/bench_pipeline.py --utts 50000 --precisions f32x,bf16 --paths sharded,stream > $out/r5za_ark.json 2>$out/r5za_ark.err
python - <<PY
import json
d=json.load(open("$out/r5za_ark.json"))
for k,v in d["runs"].items(): print(k, v.get("loop_utts_per_s"), v.get("loop_seconds"), v.get("consumer_thread_seconds"), v.get("error","")[:300])
PY
