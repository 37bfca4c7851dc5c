#!/bin/bash
# Next measurement of the 64-channel persistent f32x kernel (written at the end of round 4, not run: the round's GPU
# budget was spent).  One gpurun call:  gpurun --timeout 600 -- 'tools/r5_pers64_sq.sh'
#   1. SQ counters of grid_conv_x3_pers64_kernel (default tile loop and ASV_AMD_X3_PERS=17 = phase after phase) and of the one-tile kernel
#      (=3): matrix-pipe busy cycles, wave cycles / waits, LDS activity + bank conflicts, shader clock (GRBM_GUI_ACTIVE / 8 / duration);
#   2. the developer build's ablations of the same layer (ASV_AMD_X3_PERS_ABL: 1 no stores, 2 no row fetch / split, 4 no K loop, 8 no
#      exchange + epilogue) on the phase-after-phase form.
# tools/sq_summary.py turns each pass into a per-kernel table.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
short="--model resnet --precision f32x --streams 1 --steps 4 --warmup 2 --no-profile --min-seconds 0.05 --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic"
for pers in 1 17 3; do
  ASV_AMD_X3_PERS=$pers timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $out/r5_sq_pers$pers -- python $root/bench.py $short > /dev/null 2>&1
  cp $out/r5_sq_pers$pers/*/*counter_collection.csv $out/r5_sq_pers$pers.csv 2>/dev/null
  cp $out/r5_sq_pers$pers/*/*kernel_trace.csv $out/r5_sq_pers${pers}_trace.csv 2>/dev/null
  rm -rf $out/r5_sq_pers$pers
  python $root/tools/sq_summary.py $out/r5_sq_pers$pers.csv $out/r5_sq_pers${pers}_trace.csv $out/r5_mfma_util_pers$pers.json > /dev/null && \
    python -c "
import json
d = json.load(open('$out/r5_mfma_util_pers$pers.json'))
for k, v in d.items():
    if 'grid_conv_x3' in k: print('ASV_AMD_X3_PERS=$pers', k[:80], v)"
done
cd $root
for abl in 0 1 2 4 8 6 15; do
  echo "ASV_AMD_X3_PERS=17 ASV_AMD_X3_PERS_ABL=$abl"
  ASV_AMD_LIB=$root/asv-subtools_amd/libasv_amd_dev.so ASV_AMD_LIVE_TUNE=1 ASV_AMD_X3_PERS=17 ASV_AMD_X3_PERS_ABL=$abl timeout 200 python bench.py --model resnet --precision f32x --streams 1 --cpu-seconds 0 \
    --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --min-seconds 0.3 --per-op 2>&1 >/dev/null | grep -E "op +(27|28) tdnn_gemm"
done | tee $out/r5_pers64_abl.txt
