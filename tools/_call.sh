cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_end_to_end.py -q -x 2>&1 | tail -12 > gpurun_out/r2j_pytest.txt
python bench.py --model resnet --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2j_resnet.json 2> gpurun_out/r2j_resnet_perop.txt
tail -4 gpurun_out/r2j_pytest.txt; cut -c1-220 gpurun_out/r2j_resnet.json; grep -E "128->128|256->256" gpurun_out/r2j_resnet_perop.txt | head -4
