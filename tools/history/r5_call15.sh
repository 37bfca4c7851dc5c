#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
P=$root/asv-subtools_amd/tools_p8_probe
( for shape in "256 128 256 1 3 1" "2048 128 256 1 5 2" "2048 64 256 1 5 2" "4096 192 320 3 5 2" "4096 192 320 2 5 2" "6144 512 512 5 5 2" "66048 64 256 1 5 2" "70400 128 512 2 5 2" "131328 192 768 1 3 2"; do timeout 120 $P $shape || echo "probe rc=$?"; done ) > $out/r5p_p8_small.txt 2>&1
grep -E "rows=|differ|reference|rc=|error|supported" $out/r5p_p8_small.txt | cut -c1-200
if grep -q "HIP error\|probe rc" $out/r5p_p8_small.txt; then echo "small shapes failed: stopping"; exit 0; fi
( for shape in "52224 512 512 3 20 7" "130560 512 512 3 10 7" "77824 1024 1024 1 10 7" "77824 3072 1536 1 5 7"; do timeout 200 $P $shape || echo "probe rc=$?"; done ) > $out/r5p_p8_shapes.txt 2>&1
grep -E "rows=|differ|big3 128|p8 " $out/r5p_p8_shapes.txt | cut -c1-200
