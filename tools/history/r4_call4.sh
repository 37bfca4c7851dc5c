#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_ecapa.py tests/test_gpu_kernels.py tests/test_gpu_grid_conv_x3.py tests/test_gpu_resnet.py tests/test_gpu_full_size_parity.py -q --no-header -p no:cacheprovider > $out/r4i_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4i_pytest.txt
tail -5 $out/r4i_pytest.txt | cut -c1-300
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0"
python bench.py --model ecapa --precision f32x $one --per-op > $out/r4i_ecapa_f32x.json 2> $out/r4i_ecapa_f32x_perop.txt
tail -1 $out/r4i_ecapa_f32x.json | cut -c1-200
python bench.py --model resnet --precision f32x $one --per-op > $out/r4i_resnet_f32x.json 2> $out/r4i_resnet_f32x_perop.txt
tail -1 $out/r4i_resnet_f32x.json | cut -c1-200
