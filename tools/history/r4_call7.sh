#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
timeout 900 python -m pytest tests/test_gpu_xvector.py tests/test_gpu_devlib.py tests/test_gpu_end_to_end.py tests/test_gpu_pipeline.py -q --no-header -p no:cacheprovider > $out/r4g_pytest.txt 2>&1
echo "pytest rc=$?" >> $out/r4g_pytest.txt
tail -5 $out/r4g_pytest.txt | cut -c1-300
: > $out/r4g_small_batch_tiles.txt
for b in 1 16 64 120; do
  for mode in 1 0; do
    ASV_AMD_CHAIN_TAIL=$mode python bench.py --batch $b --streams 1 --cpu-seconds 0 --no-supplementary --eer-trials 0 --min-seconds 0.5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('batch %4d ASV_AMD_CHAIN_TAIL=$mode  %10.1f utt/s  %.4f ms/step  per_launch us %s' % ($b, d['value'], d['ms_per_step'], [p.get('us') for p in r.get('per_launch',[])]))" >> $out/r4g_small_batch_tiles.txt
  done
done
cat $out/r4g_small_batch_tiles.txt
