cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ecapa.py tests/test_gpu_full_size_parity.py tests/test_gpu_pipeline.py -q -x -k "not c4_standin_full" 2>&1 | tail -8 > gpurun_out/r2g_pytest.txt
timeout 200 python tools/res2_dbg.py 2>&1 | tail -1 > gpurun_out/r2g_res2_dbg.txt
python bench.py --model ecapa --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2g_ecapa.json 2> gpurun_out/r2g_ecapa_perop.txt
tail -3 gpurun_out/r2g_pytest.txt; cat gpurun_out/r2g_res2_dbg.txt; cut -c1-220 gpurun_out/r2g_ecapa.json; sed -n 2,14p gpurun_out/r2g_ecapa_perop.txt
