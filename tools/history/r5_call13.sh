#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ecapa.py tests/test_gpu_xvector.py tests/test_gpu_devlib.py -m gpu -q --no-header -p no:cacheprovider > $out/r5m_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r5m_pytest.txt; tail -5 $out/r5m_pytest.txt | cut -c1-300
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
done | tee $out/r5m_p8_two_phase_model_ab.txt
