#!/bin/bash
# round 4, call 13: where the 64-channel persistent f32x kernel spends its time - developer build, ablations (results are garbage)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
: > $out/r4n_pers64_abl.txt
for abl in 0 1 2 4 8 6 15; do
  echo "ASV_AMD_X3_PERS_ABL=$abl (1 no stores, 2 no row fetch / split, 4 no K loop, 8 no exchange + epilogue)" >> $out/r4n_pers64_abl.txt
  ASV_AMD_LIB=$root/asv-subtools_amd/libasv_amd_dev.so ASV_AMD_LIVE_TUNE=1 ASV_AMD_X3_PERS_ABL=$abl timeout 200 python bench.py --model resnet --precision f32x --streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --min-seconds 0.3 --per-op 2>&1 >/dev/null | grep -E "op +[0-9]+ tdnn_gemm +64->64" | head -2 >> $out/r4n_pers64_abl.txt
done
cat $out/r4n_pers64_abl.txt
