#!/bin/bash
# round 4, call 11: persistent f32x kernel for the 32-channel stage - bit identity + A/B on the ResNet (ragged C5 workload)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_grid_conv_x3.py tests/test_gpu_resnet.py -q --no-header -p no:cacheprovider -x > $out/r4l_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4l_pytest.txt
tail -15 $out/r4l_pytest.txt | cut -c1-300
: > $out/r4l_x3_pers_ab.txt
for i in 1 2; do
  for pers in 1 3 0; do
    ASV_AMD_X3_PERS=$pers python bench.py --model resnet --precision f32x --lengths 200:1000 --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --no-profile --min-seconds 1.0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('resnet f32x ASV_AMD_X3_PERS=$pers  %10.1f utt/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" >> $out/r4l_x3_pers_ab.txt
  done
done
ASV_AMD_X3_PERS=1 python bench.py --model resnet --precision f32x --streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --per-op 2>&1 >/dev/null | grep -E "op +[0-9]+ tdnn_gemm +(32->32|64->64)" | head -6 >> $out/r4l_x3_pers_ab.txt
ASV_AMD_X3_PERS=0 python bench.py --model resnet --precision f32x --streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --per-op 2>&1 >/dev/null | grep -E "op +[0-9]+ tdnn_gemm +(32->32|64->64)" | head -6 >> $out/r4l_x3_pers_ab.txt
cat $out/r4l_x3_pers_ab.txt
