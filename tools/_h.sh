cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2h
python3 tools/gpu_mstep_host_split.py > gpurun_out/r2h/split.txt 2>&1
python3 tools/gpu_step_cprofile.py > gpurun_out/r2h/cprof.txt 2>&1
cat gpurun_out/r2h/split.txt; head -60 gpurun_out/r2h/cprof.txt
