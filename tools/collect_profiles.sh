#!/bin/bash
# Collects the round's measurement artefacts on a GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh <tag>     ->  gpurun_out/<tag>_*   (copy the summaries you want judged into profiles/)
# Every rocprofv3 pass has its own short timeout: a PMC pass that needs replay can hang for the whole gpurun limit.
set -u
tag=${1:-r1}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json | cut -c1-300
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt -- python $root/bench.py --cpu-seconds 0 > /dev/null 2>&1
cp $out/${tag}_kt/*/*kernel_stats.csv $out/${tag}_bench_kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_$c -- python $root/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-profile > /dev/null 2>&1
  cp $out/${tag}_pmc_$c/*/*counter_collection.csv $out/${tag}_pmc_${c}.csv 2>/dev/null
done
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_pmc_sq -- python $root/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-profile > /dev/null 2>&1
cp $out/${tag}_pmc_sq/*/*counter_collection.csv $out/${tag}_pmc_sq.csv 2>/dev/null
cp $out/${tag}_pmc_sq/*/*kernel_trace.csv $out/${tag}_pmc_sq_trace.csv 2>/dev/null
ls -la $out | grep ${tag}_ | head -20
