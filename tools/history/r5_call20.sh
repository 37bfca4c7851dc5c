#!/bin/bash
# 8-phase kernel with a packed channel remainder (the first layer; the kernel code measured here is commit 1065696 - taken out afterwards: slower) + the sharded path without flushes at segment ends
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
tag=${1:-r5v}
cd $root
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_xvector.py tests/test_gpu_ecapa.py tests/test_gpu_pipeline.py -m gpu -q --no-header -p no:cacheprovider -x > $out/${tag}_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/${tag}_pytest.txt; tail -8 $out/${tag}_pytest.txt | cut -c1-400
if ! grep -q "pytest rc=0" $out/${tag}_pytest.txt; then echo "tests failed: stopping"; exit 0; fi
cd /tmp && export TMPDIR=/tmp
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic --gate-seeds 0"
two="--cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic --gate-seeds 0 --no-profile"
for rep in 1 2; do
for tail in 1 0; do
  for m in "xvector bf16" "ecapa bf16"; do
    set -- $m
    ASV_AMD_P8_TAIL=$tail timeout 300 python $root/bench.py --model $1 --precision $2 $one 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('one stream  P8_TAIL=$tail $1 $2 value', d['value'], 'ms/step', d['ms_per_step'], 'gemm frac', r.get('frac'), [(x['layer'], x['us'], x['tflops']) for x in r.get('per_launch', [])][:1])"
    ASV_AMD_P8_TAIL=$tail timeout 300 python $root/bench.py --model $1 --precision $2 $two 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('two streams P8_TAIL=$tail $1 $2 value', d['value'], 'ms/step', d['ms_per_step'])"
  done
done
done | tee $out/${tag}_p8_tail_model_ab.txt
cd $root
for rep in 1 2; do
timeout 300 python tools/bench_pipeline.py --utts 50000 --precisions f32x,bf16 --paths stream,sharded > $out/${tag}_ark_$rep.json 2>$out/${tag}_ark_$rep.err
python - <<PY
import json
d=json.load(open("$out/${tag}_ark_$rep.json"))
for k,v in d["runs"].items(): print(k, v.get("loop_utts_per_s"), v.get("end_to_end_seconds"), v.get("error","")[:300])
PY
done
