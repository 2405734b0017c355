cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2m
DGPAMD_MEGA_CAP=12 python3 tools/gpu_mega_trace.py 2000 12 > gpurun_out/r2m/trace_b12_cap12.txt 2>&1
DGPAMD_MEGA_CAP=12 python3 tools/gpu_mega_trace.py 2000 6 inv > gpurun_out/r2m/trace_b6inv_cap12.txt 2>&1
cat gpurun_out/r2m/trace_b12_cap12.txt
