timeout 300 python -m pytest tests/test_gpu_fbank.py -x -q 2>&1 | tail -2; timeout 100 python tools/bench_fbank.py --cmn 0 --iters 1500 | tail -1; cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/fb -o fb -- python /root/repo/tools/bench_fbank.py --iters 1500 >/dev/null 2>&1; head -3 /root/repo/gpurun_out/fb/*kernel_stats.csv | cut -c1-160; python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open('/root/repo/gpurun_out/fb/fb_kernel_trace.csv')) if 'fbank512' in r['Kernel_Name']]
d=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
print(len(d), 'first100', sum(d[:100])/100, 'last500', sum(d[-500:])/500)
PY
