#!/bin/bash
# round 4, third GPU call (the second one, re-run after its fixes): gate statistics per model, f32x range guard, the f32 input convolution, narrow frames-domain layers on the
# f32x conv kernel (ECAPA), per-op tables, and the default bench line with its new parity_grade leg (wall time)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests/test_gpu_eer_gate.py tests/test_gpu_xvector.py tests/test_gpu_resnet.py tests/test_gpu_ecapa.py tests/test_gpu_kernels.py tests/test_gpu_grid_conv_x3.py -q --no-header -p no:cacheprovider --durations=8 > $out/r4c_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4c_pytest.txt
grep "gate table\|passed\|failed\|rc=" $out/r4c_pytest.txt | cut -c1-400
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0"
python bench.py --model resnet --precision f32x $one --per-op > $out/r4c_resnet_f32x.json 2> $out/r4c_resnet_f32x_perop.txt
tail -1 $out/r4c_resnet_f32x.json | cut -c1-200
python bench.py --model ecapa --precision f32x $one --per-op > $out/r4c_ecapa_f32x.json 2> $out/r4c_ecapa_f32x_perop.txt
tail -1 $out/r4c_ecapa_f32x.json | cut -c1-200
python bench.py --model ecapa --precision bf16 $one --per-op > $out/r4c_ecapa.json 2> $out/r4c_ecapa_perop.txt
tail -1 $out/r4c_ecapa.json | cut -c1-200
t0=$(date +%s)
python bench.py > $out/r4c_bench.json 2> $out/r4c_bench.err
echo "bench.py wall seconds: $(( $(date +%s) - t0 ))" | tee $out/r4c_bench_wall.txt
tail -1 $out/r4c_bench.json | cut -c1-600
