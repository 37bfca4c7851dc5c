cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_xvector.py tests/test_gpu_ecapa.py tests/test_gpu_full_size_parity.py tests/test_gpu_eer_gate.py -q -x 2>&1 | tail -6 > gpurun_out/r2p_pytest.txt
python bench.py --precision f32x --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2p_xvector_f32x.json 2> gpurun_out/r2p_xvector_f32x_perop.txt
python bench.py --precision f32x --model ecapa --cpu-seconds 0 --no-supplementary > gpurun_out/r2p_ecapa_f32x.json 2> /dev/null
tail -3 gpurun_out/r2p_pytest.txt; cut -c1-230 gpurun_out/r2p_xvector_f32x.json; cat gpurun_out/r2p_xvector_f32x_perop.txt | tail -8; cut -c1-230 gpurun_out/r2p_ecapa_f32x.json
