#!/bin/bash
# round 4, first GPU call: the new f32x grid-convolution kernel (parity + rate), RCCL at world size 1, developer-build tests,
# r2-final vs HEAD A/B
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_grid_conv_x3.py tests/test_gpu_resnet.py tests/test_gpu_rccl.py tests/test_gpu_devlib.py tests/test_gpu_kernels.py tests/test_gpu_xvector.py -q --no-header -p no:cacheprovider > $out/r4a_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4a_pytest.txt
tail -15 $out/r4a_pytest.txt | cut -c1-300
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0"
python bench.py --model resnet --precision f32x $one --per-op > $out/r4a_resnet_f32x.json 2> $out/r4a_resnet_f32x_perop.txt
tail -1 $out/r4a_resnet_f32x.json | cut -c1-300
python bench.py --model resnet --precision f32x $one --lengths 200:1000 > $out/r4a_resnet_f32x_ragged.json 2> $out/r4a_resnet_f32x_ragged.err
tail -1 $out/r4a_resnet_f32x_ragged.json | cut -c1-300
timeout 600 tools/r4_ab_r2lib.sh 3
