#!/bin/bash
# last check of the round: the pipeline / RCCL GPU tests on the final python side + smoke
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_pipeline.py -m gpu -q --no-header -p no:cacheprovider -x -k "rccl or sharded or device_sets" > $out/r5zc_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r5zc_pytest.txt; tail -5 $out/r5zc_pytest.txt | cut -c1-300
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
