#!/bin/bash
# round 4, call 12: the 64-channel persistent f32x kernel as 32-row tiles / two workgroups per CU (ASV_AMD_X3_PERS=1) against 64-row
# tiles / one workgroup (=5), both against the one-tile kernel (=3: 32-channel form only)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python -m pytest tests/test_gpu_grid_conv_x3.py tests/test_gpu_resnet.py -q --no-header -p no:cacheprovider -x > $out/r4m_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4m_pytest.txt
tail -6 $out/r4m_pytest.txt | cut -c1-300
: > $out/r4m_x3_pers64_ab.txt
for pers in 1 5 3; do
  echo "ASV_AMD_X3_PERS=$pers" >> $out/r4m_x3_pers64_ab.txt
  ASV_AMD_X3_PERS=$pers timeout 300 python bench.py --model resnet --precision f32x --streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --per-op 2>&1 >/dev/null | grep -E "op +[0-9]+ tdnn_gemm +64->64" | head -3 >> $out/r4m_x3_pers64_ab.txt
  ASV_AMD_X3_PERS=$pers timeout 300 python bench.py --model resnet --precision f32x --lengths 200:1000 --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --no-profile --min-seconds 1.0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('resnet f32x ragged  %10.1f utt/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" >> $out/r4m_x3_pers64_ab.txt
done
cat $out/r4m_x3_pers64_ab.txt
