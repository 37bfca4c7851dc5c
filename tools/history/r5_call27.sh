#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 100 python tools/bench_pipeline.py --utts 20000 --lengths 200,1000 --precisions f32x --paths sharded,stream --once > $out/r5ze_ark_ragged_rev.json 2>/dev/null
python - <<PY
import json
d=json.load(open("$out/r5ze_ark_ragged_rev.json"))
for v in d["all_runs"]: print(v["run"], v.get("loop_utts_per_s"), v.get("complete"), v.get("consumer_thread_seconds"), v.get("error","")[:300])
PY
