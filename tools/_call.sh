cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r2d_pytest1.txt
python bench.py --model ecapa --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2d_ecapa.json 2> gpurun_out/r2d_ecapa_perop.txt
python bench.py --model resnet --cpu-seconds 0 --no-supplementary --per-op > gpurun_out/r2d_resnet.json 2> gpurun_out/r2d_resnet_perop.txt
tail -4 gpurun_out/r2d_pytest1.txt; cut -c1-200 gpurun_out/r2d_ecapa.json; cut -c1-200 gpurun_out/r2d_resnet.json; head -12 gpurun_out/r2d_ecapa_perop.txt; head -5 gpurun_out/r2d_resnet_perop.txt
