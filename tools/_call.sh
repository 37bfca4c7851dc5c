cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 800 python -m pytest tests/test_gpu_ecapa.py tests/test_gpu_resnet.py tests/test_gpu_kernels.py tests/test_gpu_xvector.py tests/test_gpu_full_size_parity.py -q -x 2>&1 | tail -6 > gpurun_out/r2m_pytest.txt
python bench.py --model ecapa --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2m_ecapa.json 2> gpurun_out/r2m_ecapa_perop.txt
python bench.py --model resnet --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2m_resnet.json 2> gpurun_out/r2m_resnet_perop.txt
tail -3 gpurun_out/r2m_pytest.txt; cut -c1-200 gpurun_out/r2m_ecapa.json; grep -E "1536->128|128->1536|attentive" gpurun_out/r2m_ecapa_perop.txt; cut -c1-200 gpurun_out/r2m_resnet.json; grep -E "576->128|grid_gather" gpurun_out/r2m_resnet_perop.txt | head -8
