#!/bin/bash
# Round-3 measurement set on a GPU box (through gpurun from the repo root):  tools/r3_collect.sh <tag> [pmc]
#   gpurun_out/<tag>_pytest.txt / _smoke.txt   the whole GPU test suite and smoke() on this box
#   gpurun_out/<tag>_bench.json                the default bench line (two streams; supplementary records + cpu_baseline)
#   gpurun_out/<tag>_{xvector,xvector_f32x,ecapa,resnet}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the single-stream command per model
#   gpurun_out/<tag>_pmc_*.csv                 (with "pmc") FETCH_SIZE / WRITE_SIZE / SQ passes, each its own run
set -u
tag=${1:-r3}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
( cd $root && timeout 1500 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider > $out/${tag}_pytest.txt 2>&1; echo "pytest rc=$?" >> $out/${tag}_pytest.txt; tail -4 $out/${tag}_pytest.txt | cut -c1-200 )
( cd $root && timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.txt 2>&1; tail -2 $out/${tag}_smoke.txt )
cd /tmp && export TMPDIR=/tmp
python $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json | cut -c1-400
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0"
for m in "xvector bf16" "xvector f32x" "ecapa bf16" "resnet bf16"; do
  set -- $m; model=$1; prec=$2; name=$model; [ "$prec" != bf16 ] && name=${model}_$prec
  python $root/bench.py --model $model --precision $prec $one --per-op > $out/${tag}_${name}.json 2> $out/${tag}_${name}_perop.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt_$name -- python $root/bench.py --model $model --precision $prec $one > /dev/null 2>&1
  cp $out/${tag}_kt_$name/*/*kernel_stats.csv $out/${tag}_${name}_kernel_stats.csv 2>/dev/null
  rm -rf $out/${tag}_kt_$name
done
if [ "${2:-}" = "pmc" ]; then
  short="--steps 4 --warmup 2 --no-profile --min-seconds 0.05 $one"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_$c -- python $root/bench.py $short > $out/${tag}_pmc_$c.log 2>&1
    echo "pmc $c rc=$?"
    cp $out/${tag}_pmc_$c/*/*counter_collection.csv $out/${tag}_pmc_${c}.csv 2>/dev/null || echo "pmc $c: no counter file (see ${tag}_pmc_$c.log; the passes also run from a call of their own: the last lines of this script)"
    rm -rf $out/${tag}_pmc_$c
  done
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_pmc_sq -- python $root/bench.py $short > /dev/null 2>&1
  cp $out/${tag}_pmc_sq/*/*counter_collection.csv $out/${tag}_pmc_sq.csv 2>/dev/null
  cp $out/${tag}_pmc_sq/*/*kernel_trace.csv $out/${tag}_pmc_sq_trace.csv 2>/dev/null
  rm -rf $out/${tag}_pmc_sq
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_pmc_sq_f32x -- python $root/bench.py --precision f32x $short > /dev/null 2>&1
  cp $out/${tag}_pmc_sq_f32x/*/*counter_collection.csv $out/${tag}_pmc_sq_f32x.csv 2>/dev/null
  cp $out/${tag}_pmc_sq_f32x/*/*kernel_trace.csv $out/${tag}_pmc_sq_f32x_trace.csv 2>/dev/null
  rm -rf $out/${tag}_pmc_sq_f32x
fi
ls -la $out | grep ${tag}_ | awk '{print $5, $9}' | head -40
