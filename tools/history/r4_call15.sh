#!/bin/bash
# round 4, call 15: 64-channel persistent f32x kernel with the non-matrix phases inside the K loop (default) against phase after phase (=17)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 150 python -m pytest tests/test_gpu_grid_conv_x3.py tests/test_gpu_resnet.py -q --no-header -p no:cacheprovider -x > $out/r4p_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4p_pytest.txt
tail -4 $out/r4p_pytest.txt | cut -c1-300
: > $out/r4p_pers64_pipe_ab.txt
for pers in 1 17; do
  echo "ASV_AMD_X3_PERS=$pers" >> $out/r4p_pers64_pipe_ab.txt
  ASV_AMD_X3_PERS=$pers timeout 100 python bench.py --model resnet --precision f32x --streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --min-seconds 0.3 --per-op 2>&1 >/dev/null | grep -E "op +(22|27|28) tdnn_gemm" >> $out/r4p_pers64_pipe_ab.txt
done
cat $out/r4p_pers64_pipe_ab.txt
