#!/bin/bash
# "driver-run kernel times regressed 5-10 % r2 -> r3 and nobody has ruled out a real regression".
# Same-box alternating A/B of the ROUND-2-FINAL tree (git archive 93a47a4 -> build/r2tree, its own csrc built to its own
# libasv_amd.so, its own bench.py) against HEAD, x-vector bf16 single stream, per-launch microseconds of tdnn1 / tdnn2 / chain
# from each tree's own hipEvents.   usage (GPU box, repo root): tools/r4_ab_r2lib.sh [rounds]
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r4_r2lib_vs_head.txt
rounds=${1:-3}
mkdir -p $root/gpurun_out
: > $out
run() {   # label, tree, extra flags
  local line=$(cd $2 && timeout 200 python bench.py --cpu-seconds 0 --no-supplementary --min-seconds 1.0 $3 2>/dev/null | tail -1)
  python - "$1" "$line" >> $out <<'PY'
import json, sys
label, line = sys.argv[1:3]
try:
    d = json.loads(line)
    r = d.get("roofline", {})
    v = d.get("value_single_stream", d["value"])
    print("%-6s %10.1f utt/s  gemm frac %s  per_launch us %s" % (label, v, r.get("frac"), [p.get("us") for p in r.get("per_launch", [])]))
except Exception as e:
    print(label, "FAILED", e, line[:300])
PY
}
for i in $(seq $rounds); do
  run r2 $root/build/r2tree ""
  run head $root "--streams 1 --eer-trials 0"
done
cat $out
