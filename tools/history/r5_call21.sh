#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
P=$root/asv-subtools_amd/tools_p8_probe
( for shape in "52224 80 512 5 20 5" "52224 64 512 5 20 5" "52224 128 512 5 20 5" "77824 80 1024 5 10 5" "130560 80 512 5 10 5"; do timeout 120 $P $shape || echo "probe rc=$?"; done ) > $out/r5w_p8_tail_probe.txt 2>&1
cut -c1-200 $out/r5w_p8_tail_probe.txt | grep -v "^$"
