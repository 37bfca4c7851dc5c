#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=8 > $out/r5g_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r5g_pytest.txt; tail -14 $out/r5g_pytest.txt | cut -c1-300
