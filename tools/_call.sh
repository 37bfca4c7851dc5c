cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_xvector.py tests/test_gpu_full_size_parity.py tests/test_gpu_scoring.py -q -x 2>&1 | tail -6 > gpurun_out/r2r_pytest.txt
python bench.py --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2r_xvector.json 2> gpurun_out/r2r_xvector_perop.txt
tail -3 gpurun_out/r2r_pytest.txt; cut -c1-220 gpurun_out/r2r_xvector.json; cat gpurun_out/r2r_xvector_perop.txt | tail -7
