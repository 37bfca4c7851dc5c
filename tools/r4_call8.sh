#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_grid_conv_x3.py tests/test_gpu_ecapa.py tests/test_gpu_kernels.py -q --no-header -p no:cacheprovider > $out/r4h_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4h_pytest.txt
tail -5 $out/r4h_pytest.txt | cut -c1-300
: > $out/r4h_conv_tail_ab.txt
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --no-profile --min-seconds 0.6"
for i in 1 2; do
  for mode in 1 0; do
    for cfg in "bf16 200" "f32x 200" "bf16 ragged" "f32x ragged"; do
      set -- $cfg; prec=$1; shape=$2
      extra=""; [ "$shape" = ragged ] && extra="--lengths 200:1000"
      ASV_AMD_CONV_TAIL=$mode python bench.py --model resnet --precision $prec $one $extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('resnet %-5s %-7s ASV_AMD_CONV_TAIL=$mode  %9.1f utt/s  %.3f ms/step' % ('$prec', '$shape', d['value'], d['ms_per_step']))" >> $out/r4h_conv_tail_ab.txt
    done
  done
done
cat $out/r4h_conv_tail_ab.txt
python bench.py --model resnet --precision bf16 --streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --per-op > $out/r4h_resnet.json 2> $out/r4h_resnet_perop.txt
