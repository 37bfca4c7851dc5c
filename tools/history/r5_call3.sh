#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_xvector.py tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider \
  -k "range_guard or device_sets or ark_to_ark or sharded or im2col or neighbour or chain or golden or ragged" > $out/r5c_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r5c_pytest.txt; tail -6 $out/r5c_pytest.txt | cut -c1-300
timeout 500 python tools/bench_pipeline.py --utts 50000 > $out/r5c_ark_to_ark.json 2> $out/r5c_ark_to_ark.err
echo "pipeline rc=$?"; cat $out/r5c_ark_to_ark.json | cut -c1-3000
for th in 4 16; do
  ASV_AMD_READER_THREADS=$th timeout 200 python tools/bench_pipeline.py --utts 50000 --precisions f32x --paths stream,sharded --dir /tmp/asv_pipe_t$th 2>/dev/null | tee $out/r5c_ark_to_ark_threads$th.json | cut -c1-900
done
