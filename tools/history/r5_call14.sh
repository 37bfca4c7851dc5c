#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
P=$root/asv-subtools_amd/tools_p8_probe
( for shape in "77824 1024 1024 1 10 3" "77824 3072 1536 1 5 3" "52224 512 512 3 20 3"; do timeout 200 $P $shape || echo "probe rc=$?"; done ) > $out/r5o_p8_stamps.txt 2>&1
grep -E "rows=|stamps|big3 128|p8 two-phase  |rc=" $out/r5o_p8_stamps.txt | cut -c1-220
