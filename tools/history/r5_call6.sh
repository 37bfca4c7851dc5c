#!/bin/bash
# round 5, call 6: the whole GPU suite with the 8-phase kernel dispatched + A/B of the ECAPA / x-vector steps with it on and off
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --durations=8 > $out/r5f_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r5f_pytest.txt; tail -14 $out/r5f_pytest.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic --gate-seeds 0"
for p8 in 1 0 1 0; do
  for m in "ecapa bf16" "xvector bf16"; do
    set -- $m
    ASV_AMD_P8=$p8 python $root/bench.py --model $1 --precision $2 $one 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('P8=$p8 $1 $2 value', d['value'], 'ms/step', d['ms_per_step'], 'gemm frac', r.get('frac'), 'dominant', r.get('dominant_tflops'))"
  done
done | tee $out/r5f_p8_ab.txt
ASV_AMD_P8=1 python $root/bench.py --model ecapa --precision bf16 $one --per-op > /dev/null 2> $out/r5f_ecapa_perop.txt; grep -E "tdnn_gemm|res2" $out/r5f_ecapa_perop.txt | head -40
