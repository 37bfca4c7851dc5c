cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_xvector.py tests/test_gpu_fbank.py tests/test_gpu_resnet.py -q -x 2>&1 | tail -15 > gpurun_out/r2b_pytest1.txt
timeout 700 python -m pytest tests/test_gpu_pipeline.py -q -x -k c4_standin_full_size 2>&1 | tail -60 > gpurun_out/r2b_pytest_c4.txt
timeout 200 python tools/chain_ab.py ASV_AMD_CHAIN_POOLV 0 1 --rounds 5 > gpurun_out/r2b_ab_poolv.txt 2>&1
timeout 100 python tools/chain_dbg.py 2>&1 | tail -4 > gpurun_out/r2b_chain_dbg.txt
python bench.py --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2b_xvector.json 2> gpurun_out/r2b_xvector_perop.txt
python bench.py --model resnet --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2b_resnet.json 2> gpurun_out/r2b_resnet_perop.txt
tail -3 gpurun_out/r2b_pytest1.txt; tail -5 gpurun_out/r2b_pytest_c4.txt; cat gpurun_out/r2b_ab_poolv.txt | tail -3; cat gpurun_out/r2b_chain_dbg.txt | tail -2; cut -c1-200 gpurun_out/r2b_xvector.json; head -8 gpurun_out/r2b_resnet_perop.txt
