cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 800 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_end_to_end.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -6 > gpurun_out/r2n_pytest.txt
python bench.py --model resnet --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2n_resnet.json 2> gpurun_out/r2n_resnet_perop.txt
tail -3 gpurun_out/r2n_pytest.txt; cut -c1-200 gpurun_out/r2n_resnet.json; sed -n 20,28p gpurun_out/r2n_resnet_perop.txt; grep -E "grid_gather|taps=\[-2[12]" gpurun_out/r2n_resnet_perop.txt | head -8
