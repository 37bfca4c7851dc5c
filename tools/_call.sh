cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r2_pytest.txt
tail -2 gpurun_out/r2_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.txt 2>&1; tail -2 gpurun_out/r2_smoke.txt
timeout 800 tools/r2_collect.sh r2 pmc
