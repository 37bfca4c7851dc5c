cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r2_pytest.txt
tail -2 gpurun_out/r2_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.txt 2>&1; tail -3 gpurun_out/r2_smoke.txt
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/r2_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2_bench.err
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_kt -- python $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --no-supplementary > /dev/null 2>&1
cp $GRAFT_REPO_ROOT/gpurun_out/r2_kt/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/r2_xvector_kernel_stats.csv; rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2_kt
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r2_bench.json | cut -c1-200
