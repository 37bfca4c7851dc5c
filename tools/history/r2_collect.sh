#!/bin/bash
# Round-2 measurement set on a GPU box (through gpurun from the repo root):  tools/r2_collect.sh <tag> [pmc]
#   gpurun_out/<tag>_bench.json            the default bench line (with supplementary records + cpu_baseline)
#   gpurun_out/<tag>_{xvector,ecapa,resnet}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command per model
#   gpurun_out/<tag>_pmc_*.csv             (with "pmc") FETCH_SIZE / WRITE_SIZE / SQ passes, each its own run
set -u
tag=${1:-r2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json | cut -c1-600
for m in xvector ecapa resnet; do
  python $root/bench.py --model $m --cpu-seconds 0 --no-supplementary --per-op > $out/${tag}_${m}.json 2> $out/${tag}_${m}_perop.txt
  timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt_$m -- python $root/bench.py --model $m --cpu-seconds 0 --no-supplementary > /dev/null 2>&1
  cp $out/${tag}_kt_$m/*/*kernel_stats.csv $out/${tag}_${m}_kernel_stats.csv 2>/dev/null
  rm -rf $out/${tag}_kt_$m
done
if [ "${2:-}" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_$c -- python $root/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-profile --no-supplementary --min-seconds 0.05 > /dev/null 2>&1
    cp $out/${tag}_pmc_$c/*/*counter_collection.csv $out/${tag}_pmc_${c}.csv 2>/dev/null
    rm -rf $out/${tag}_pmc_$c
  done
  timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_pmc_sq -- python $root/bench.py --steps 4 --warmup 2 --cpu-seconds 0 --no-profile --no-supplementary --min-seconds 0.05 > /dev/null 2>&1
  cp $out/${tag}_pmc_sq/*/*counter_collection.csv $out/${tag}_pmc_sq.csv 2>/dev/null
  cp $out/${tag}_pmc_sq/*/*kernel_trace.csv $out/${tag}_pmc_sq_trace.csv 2>/dev/null
  rm -rf $out/${tag}_pmc_sq
fi
ls -la $out | grep ${tag}_ | head -30
