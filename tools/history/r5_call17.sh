#!/bin/bash
# f32x 8-phase kernel (kernels_tdnn_p8x.hip): parity tests, then model-level A/B (ASV_AMD_P8X=1 / 0) on the f32x benches
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
tag=${1:-r5s}
cd $root
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ecapa.py -m gpu -q --no-header -p no:cacheprovider -x -k "p8x or 8phase or f32x" > $out/${tag}_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/${tag}_pytest.txt; tail -15 $out/${tag}_pytest.txt | cut -c1-400
if ! grep -q "pytest rc=0" $out/${tag}_pytest.txt; then echo "tests failed: stopping"; exit 0; fi
cd /tmp && export TMPDIR=/tmp
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic --gate-seeds 0"
two="--cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic --gate-seeds 0 --no-profile"
for rep in 1 2; do
for p8 in 1 0; do
  for m in "ecapa f32x" "xvector f32x"; do
    set -- $m
    ASV_AMD_P8X=$p8 timeout 300 python $root/bench.py --model $1 --precision $2 $one 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('one stream  P8X=$p8 $1 $2 value', d['value'], 'ms/step', d['ms_per_step'], 'gemm frac', r.get('frac'), 'dominant', r.get('dominant_kernel'), r.get('dominant_tflops'))"
    ASV_AMD_P8X=$p8 timeout 300 python $root/bench.py --model $1 --precision $2 $two 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('two streams P8X=$p8 $1 $2 value', d['value'], 'ms/step', d['ms_per_step'])"
  done
done
done | tee $out/${tag}_p8x_model_ab.txt
ASV_AMD_P8X=1 timeout 300 python $root/bench.py --model ecapa --precision f32x $one --per-op 2>$out/${tag}_ecapa_f32x_perop_p8x.txt >/dev/null
ASV_AMD_P8X=0 timeout 300 python $root/bench.py --model ecapa --precision f32x $one --per-op 2>$out/${tag}_ecapa_f32x_perop_x3.txt >/dev/null
grep -i "tdnn\|p8x\|x3" $out/${tag}_ecapa_f32x_perop_p8x.txt | head -30
