for d in ${DIAGS:-0 4 8 12 1 9}; do
  echo "## DGPAMD_JSEP_DIAG=$d (1: all mixed, 4: all class 1, 8: no staging after the first step)"
  DGPAMD_JSEP_DIAG=$d ORDER=1 CHECK=0 python tools/gpu_linkgp_bench.py matern2.5 2000 5 0 4096 2>&1 | grep -v amdgpu.ids
done
