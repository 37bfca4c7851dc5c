#!/bin/bash
# Developer tool: A/B of two builds of the library on ONE box (box-to-box variance is +-4 %): bench.py alternately with the
# in-tree libasv_amd.so and with $1 (ASV_AMD_LIB), x-vector twice each, ECAPA and ResNet once each.  usage: tools/lib_ab.sh <other.so> <tag>
other=$(readlink -f "$1"); tag=${2:-ab}
mkdir -p gpurun_out
out=gpurun_out/${tag}_lib_ab.txt
: > $out
run() {   # label, lib ('' = default), model
  local t0=$(date +%s.%N)
  local line=$(ASV_AMD_LIB=$2 timeout 120 python bench.py --model $3 --no-supplementary --cpu-seconds 0 --min-seconds 0.6 2>/dev/null | tail -1)
  python - "$1" "$3" "$line" >> $out <<'PY'
import json, sys
label, model, line = sys.argv[1:4]
try:
    d = json.loads(line)
    r = d.get("roofline", {})
    print("%-8s %-8s %10.1f utt/s  %.4f ms/step  gemm frac %s  per_launch %s" % (label, model, d["value"], d["ms_per_step"], r.get("frac"),
          [(p["us"], p["tflops"]) for p in r.get("per_launch", [])]))
except Exception as e:
    print(label, model, "FAILED", e, line[:200])
PY
}
run base "" xvector
run other "$other" xvector
run base "" xvector
run other "$other" xvector
run base "" ecapa
run other "$other" ecapa
run base "" resnet
run other "$other" resnet
cat $out
