timeout 300 python -m pytest tests/test_gpu_fbank.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/fb -o fb -- python /root/repo/tools/bench_fbank.py --iters 1500 >/dev/null 2>&1; head -3 /root/repo/gpurun_out/fb/*kernel_stats.csv | cut -c1-160
