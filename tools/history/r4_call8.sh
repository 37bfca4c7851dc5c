#!/bin/bash
# round 4, call 8: the strided gather with the elementwise prologue - bit identity + A/B on the ResNet (ragged C5 workload)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_ecapa.py tests/test_gpu_grid_conv_x3.py -q --no-header -p no:cacheprovider > $out/r4i_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4i_pytest.txt
tail -5 $out/r4i_pytest.txt | cut -c1-300
: > $out/r4i_gather_fuse_ab.txt
for i in 1 2; do
  for prec in bf16 f32x; do
    for nf in 0 1; do
      ASV_AMD_NO_GATHER_FUSE=$nf python bench.py --model resnet --precision $prec --lengths 200:1000 --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --no-profile --min-seconds 1.0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('resnet $prec ASV_AMD_NO_GATHER_FUSE=$nf  %10.1f utt/s  %.4f ms/step  batch %s' % (d['value'], d['ms_per_step'], d['config'].get('batch')))" >> $out/r4i_gather_fuse_ab.txt
    done
  done
done
cat $out/r4i_gather_fuse_ab.txt
