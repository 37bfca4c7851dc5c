#!/bin/bash
# round 5, call 4: first run of the 8-phase kernel (tools_p8_probe: bitwise against big3, f64 samples, interleaved timing) + the
# consumer-thread timing of the f32x stream path
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
P=$root/asv-subtools_amd/tools_p8_probe
( for shape in "2048 128 256 1 5 2" "2048 64 256 1 5 2" "4096 192 320 3 5 2" "6144 512 512 5 5 2"; do timeout 120 $P $shape || echo "probe rc=$?"; done ) > $out/r5d_p8_small.txt 2>&1
tail -40 $out/r5d_p8_small.txt | cut -c1-200
if grep -q "HIP error\|probe rc" $out/r5d_p8_small.txt; then echo "small shapes failed: stopping"; exit 0; fi
( for shape in "52224 512 512 3 20 7" "130560 512 512 3 10 7" "77824 1024 1024 1 10 7" "77824 3072 1536 1 5 7" "52224 512 1536 1 20 7"; do timeout 200 $P $shape || echo "probe rc=$?"; done ) > $out/r5d_p8_shapes.txt 2>&1
cat $out/r5d_p8_shapes.txt | cut -c1-200
timeout 200 python tools/bench_pipeline.py --utts 50000 --precisions f32x,bf16 --paths stream > $out/r5d_ark_stream.json 2>/dev/null; cat $out/r5d_ark_stream.json | cut -c1-1500
