#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
two="--cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic --gate-seeds 0 --no-profile"
for rep in 1 2 3; do
for p8 in 1 0; do
  for m in "xvector bf16" "ecapa bf16"; do
    set -- $m
    ASV_AMD_P8=$p8 python $root/bench.py --model $1 --precision $2 $two 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('two streams P8=$p8 $1 $2 value', d['value'], 'ms/step', d['ms_per_step'], 'single', d.get('value_single_stream'))"
  done
done
done | tee $out/r5k_p8_two_streams_ab.txt
cd $root && timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --no-header -p no:cacheprovider -k "range_guard or device_sets or ark_to_ark or sharded" 2>&1 | tail -3
