#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 300 python tools/r5_diag_guard.py > $out/r5b_diag_guard.txt 2>&1; tail -30 $out/r5b_diag_guard.txt | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py tests/test_gpu_rccl.py -m gpu -q --no-header -p no:cacheprovider \
  -k "device_sets or im2col or rccl" > $out/r5b_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r5b_pytest.txt; tail -5 $out/r5b_pytest.txt | cut -c1-300
