#!/bin/bash
# round 5, call 1: the new script paths on a device - range guard through the script (ark / scp / sharded), DeviceSets, the C-ABI
# im2col default - then the ark -> ark rates of the round-4/5 loaders (VERDICT r4 missing item 1) and a short bench line (flat dominant fields)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_kernels.py tests/test_gpu_xvector.py tests/test_gpu_rccl.py -m gpu -q --no-header -p no:cacheprovider -x \
  -k "range_guard or device_sets or ark_to_ark or sharded or im2col or rccl" > $out/r5a_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r5a_pytest.txt; tail -5 $out/r5a_pytest.txt | cut -c1-300
timeout 500 python tools/bench_pipeline.py --utts 50000 > $out/r5a_ark_to_ark.json 2> $out/r5a_ark_to_ark.err
echo "pipeline rc=$?"; cat $out/r5a_ark_to_ark.json | cut -c1-3000; tail -3 $out/r5a_ark_to_ark.err
for th in 1 2 8; do
  ASV_AMD_READER_THREADS=$th timeout 200 python tools/bench_pipeline.py --utts 30000 --precisions bf16 --paths stream,sharded --dir /tmp/asv_pipe_t$th 2>/dev/null | tee $out/r5a_ark_to_ark_threads$th.json | cut -c1-900
done
cd /tmp && export TMPDIR=/tmp
timeout 300 python $root/bench.py --no-supplementary --cpu-seconds 0 --no-traffic --eer-trials 0 > $out/r5a_bench_short.json 2> $out/r5a_bench_short.err
tail -1 $out/r5a_bench_short.json | cut -c1-2500
nproc; free -g | head -2
