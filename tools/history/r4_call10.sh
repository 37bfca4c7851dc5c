#!/bin/bash
# round 4, call 10: engines / streams in flight at configs[1] (256 utterances per step), bf16 and f32x
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
: > $out/r4k_streams.txt
for i in 1 2; do
  for prec in bf16 f32x; do
    for st in 1 2 3 4; do
      python bench.py --precision $prec --streams $st --cpu-seconds 0 --no-supplementary --eer-trials 0 --gate-seeds 0 --no-traffic --no-profile --min-seconds 1.0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('xvector $prec b256 streams $st  %10.1f utt/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" >> $out/r4k_streams.txt
    done
  done
done
cat $out/r4k_streams.txt
