#!/bin/bash
# Round-4 measurement set on a GPU box (through gpurun from the repo root):  tools/r4_collect.sh <tag> [pmc]
#   gpurun_out/<tag>_pytest.txt / _smoke.txt   the whole GPU test suite and smoke() on this box
#   gpurun_out/<tag>_bench.json                the default bench line (configs[1] at 256 utterances, two streams; parity_grade tables,
#                                              supplementary records, cpu_baseline, roofline.traffic from its own rocprofv3 passes)
#   gpurun_out/<tag>_{xvector,xvector_f32x,ecapa,ecapa_f32x,resnet,resnet_f32x}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the
#                                              single-stream command per model / mode, + the per-op tables
#   gpurun_out/<tag>_pmc_*.csv                 (with "pmc") FETCH_SIZE / WRITE_SIZE / SQ passes, each its own run
set -u
tag=${1:-r4}
do_pmc=${2:-}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
( cd $root && timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=12 > $out/${tag}_pytest.txt 2>&1; echo "pytest rc=$?" >> $out/${tag}_pytest.txt; tail -4 $out/${tag}_pytest.txt | cut -c1-200 )
( cd $root && timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.txt 2>&1; tail -2 $out/${tag}_smoke.txt )
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
python $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "bench.py wall seconds: $(( $(date +%s) - t0 ))" | tee $out/${tag}_bench_wall.txt
tail -1 $out/${tag}_bench.json | cut -c1-400
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic"
for m in "xvector bf16" "xvector f32x" "ecapa bf16" "ecapa f32x" "resnet bf16" "resnet f32x"; do
  set -- $m; model=$1; prec=$2; name=$model; [ "$prec" != bf16 ] && name=${model}_$prec
  python $root/bench.py --model $model --precision $prec $one --per-op > $out/${tag}_${name}.json 2> $out/${tag}_${name}_perop.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt_$name -- python $root/bench.py --model $model --precision $prec $one > /dev/null 2>&1
  cp $out/${tag}_kt_$name/*/*kernel_stats.csv $out/${tag}_${name}_kernel_stats.csv 2>/dev/null
  rm -rf $out/${tag}_kt_$name
done
if [ "$do_pmc" = "pmc" ]; then
  short="--steps 4 --warmup 2 --no-profile --min-seconds 0.05 $one"
  for cfg in "xvector bf16" "resnet f32x"; do
    set -- $cfg; model=$1; prec=$2
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_${model}_$c -- python $root/bench.py --model $model --precision $prec $short > $out/${tag}_pmc_${model}_$c.log 2>&1
      echo "pmc $model $c rc=$?"
      cp $out/${tag}_pmc_${model}_$c/*/*counter_collection.csv $out/${tag}_pmc_${model}_${c}.csv 2>/dev/null
      rm -rf $out/${tag}_pmc_${model}_$c
    done
  done
  for cfg in "xvector bf16" "xvector f32x" "resnet f32x"; do
    set -- $cfg; model=$1; prec=$2
    timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_pmc_sq_${model}_$prec -- python $root/bench.py --model $model --precision $prec $short > /dev/null 2>&1
    cp $out/${tag}_pmc_sq_${model}_$prec/*/*counter_collection.csv $out/${tag}_pmc_sq_${model}_$prec.csv 2>/dev/null
    cp $out/${tag}_pmc_sq_${model}_$prec/*/*kernel_trace.csv $out/${tag}_pmc_sq_${model}_${prec}_trace.csv 2>/dev/null
    rm -rf $out/${tag}_pmc_sq_${model}_$prec
  done
fi
ls -la $out | grep ${tag}_ | awk '{print $5, $9}' | head -60
