#!/bin/bash
# round 4, call 9: ResNetXvector with the frame-weighting poolings (grid_flatten + sequence domain) on the device
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_resnet.py tests/test_gpu_xvector.py -q --no-header -p no:cacheprovider > $out/r4j_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4j_pytest.txt
tail -30 $out/r4j_pytest.txt | cut -c1-400
