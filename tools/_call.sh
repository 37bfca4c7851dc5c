cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 60 asv-subtools_amd/tools_coissue_probe > gpurun_out/r2o_coissue.txt 2>&1; cat gpurun_out/r2o_coissue.txt
