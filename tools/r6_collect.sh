#!/bin/bash
# Round-6 measurement set on a GPU box (through gpurun from the repo root):  tools/r6_collect.sh <tag> [pmc]
#   gpurun_out/<tag>_pytest.txt / _smoke.txt   the whole GPU test suite and smoke() on this box (smoke prints the library hash: every file
#                                              of this collection belongs to that build)
#   gpurun_out/<tag>_bench.json (+ _bench_full.json)   the default bench line as the driver reads it (compact, <= 4 KB) and the full record
#   gpurun_out/<tag>_{xvector,xvector_f32m,xvector_f32x,ecapa,ecapa_f32m,resnet,resnet_f32x}_kernel_stats.csv   rocprofv3 --kernel-trace --stats
#                                              of the single-stream command per model / mode, + the per-op tables
#   gpurun_out/<tag>_chain_stamps.txt          in-kernel s_memtime stamps (ASV_AMD_CHAIN_DBG=1): shader clock under load + phase cycles, bf16 chain and f32m chain
#   gpurun_out/<tag>_loaders.txt / .json       tools/bench_loaders.py: the host side of --sharded under 1 / 2 / 4 / 8 ranks
#   gpurun_out/<tag>_pmc_*_per_kernel.csv, <tag>_sq_*.json   (with "pmc") FETCH_SIZE / WRITE_SIZE per kernel (x-vector bf16, x-vector f32m, ECAPA bf16,
#                                              ECAPA f32m), SQ passes through tools/sq_summary.py with the stamped clocks
set -u
tag=${1:-r6}
do_pmc=${2:-}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
( cd $root && timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider --durations=12 > $out/${tag}_pytest.txt 2>&1; echo "pytest rc=$?" >> $out/${tag}_pytest.txt; tail -4 $out/${tag}_pytest.txt | cut -c1-200 )
( cd $root && timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.txt 2>&1; tail -2 $out/${tag}_smoke.txt; grep sha256 $out/${tag}_smoke.txt > $out/${tag}_library_hash.txt )
cd /tmp && export TMPDIR=/tmp
t0=$(date +%s)
python $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "bench.py wall seconds: $(( $(date +%s) - t0 ))" | tee $out/${tag}_bench_wall.txt
cp $out/bench_full.json $out/${tag}_bench_full.json 2>/dev/null
tail -1 $out/${tag}_bench.json | wc -c
tail -1 $out/${tag}_bench.json | cut -c1-300
one="--streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --no-traffic --gate-seeds 0"
for m in "xvector bf16" "xvector f32m" "xvector f32x" "ecapa bf16" "ecapa f32m" "resnet bf16" "resnet f32x"; do
  set -- $m; model=$1; prec=$2; name=$model; [ "$prec" != bf16 ] && name=${model}_$prec
  python $root/bench.py --model $model --precision $prec $one --per-op > $out/${tag}_${name}.json 2> $out/${tag}_${name}_perop.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_kt_$name -- python $root/bench.py --model $model --precision $prec $one > /dev/null 2>&1
  cp $out/${tag}_kt_$name/*/*kernel_stats.csv $out/${tag}_${name}_kernel_stats.csv 2>/dev/null
  rm -rf $out/${tag}_kt_$name
done
short="--steps 4 --warmup 2 --no-profile --min-seconds 0.05 $one"
( for prec in bf16 f32m; do echo "== $prec"; ASV_AMD_CHAIN_DBG=1 python $root/bench.py --precision $prec $short 2>&1 >/dev/null | grep "chain dbg" | tail -2 | cut -c1-400; done ) > $out/${tag}_chain_stamps.txt
cat $out/${tag}_chain_stamps.txt | cut -c1-200
( cd $root && timeout 900 python tools/bench_loaders.py --utts 50000 --ragged-utts 20000 --repeats 2 > $out/${tag}_loaders.json 2> $out/${tag}_loaders.txt; cat $out/${tag}_loaders.txt )
if [ "$do_pmc" = "pmc" ]; then
  clk_bf16=$(grep -A1 "== bf16" $out/${tag}_chain_stamps.txt | grep -o "shader clock [0-9]* MHz" | head -1 | awk '{print $3}')
  clk_f32m=$(grep -A1 "== f32m" $out/${tag}_chain_stamps.txt | grep -o "shader clock [0-9]* MHz" | head -1 | awk '{print $3}')
  sq="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  for cfg in "xvector bf16" "xvector f32m" "ecapa bf16" "ecapa f32m"; do
    set -- $cfg; model=$1; prec=$2
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/${tag}_pmc_${model}_${prec}_$c -- python $root/bench.py --model $model --precision $prec $short > /dev/null 2>&1
      echo "pmc $model $prec $c rc=$?"
      python $root/tools/pmc_per_kernel.py $out/${tag}_pmc_${model}_${prec}_$c/*/*counter_collection.csv $out/${tag}_pmc_${model}_${prec}_${c}_per_kernel.csv 2>/dev/null
      rm -rf $out/${tag}_pmc_${model}_${prec}_$c
    done
    timeout 300 rocprofv3 --pmc $sq --kernel-trace --output-format csv -d $out/${tag}_pmc_sq_${model}_$prec -- python $root/bench.py --model $model --precision $prec $short > /dev/null 2>&1
    clk=""; [ "$model" = xvector ] && [ "$prec" = bf16 ] && [ -n "$clk_bf16" ] && clk="--clock tdnn_chain_kernel=$clk_bf16"
    [ "$model" = xvector ] && [ "$prec" = f32m ] && [ -n "$clk_f32m" ] && clk="--clock tdnn_chainm_kernel=$clk_f32m"
    python $root/tools/sq_summary.py $out/${tag}_pmc_sq_${model}_$prec/*/*counter_collection.csv $out/${tag}_pmc_sq_${model}_$prec/*/*kernel_trace.csv $out/${tag}_sq_${model}_$prec.json $clk > /dev/null 2>&1
    rm -rf $out/${tag}_pmc_sq_${model}_$prec
  done
fi
ls -la $out | grep ${tag}_ | awk '{print $5, $9}' | head -80
