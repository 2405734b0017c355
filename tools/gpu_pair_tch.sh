#!/bin/bash
# Test points per workgroup of the Matern pair kernel (DGPAMD_JSEP_TCH), same box.
for rep in 1 2; do for t in 32 64 128 16; do
  echo "## TCH=$t"
  DGPAMD_JSEP_TCH=$t ORDER=1 CHECK=0 python tools/gpu_linkgp_bench.py matern2.5 2000 5 0 4096 2>&1 | grep -v amdgpu.ids
  DGPAMD_JSEP_TCH=$t ORDER=1 CHECK=0 python tools/gpu_linkgp_bench.py matern2.5 5000 10 2 512 2>&1 | grep -v amdgpu.ids
  DGPAMD_JSEP_TCH=$t ORDER=1 CHECK=0 python tools/gpu_linkgp_bench.py matern2.5 500 3 0 8192 2>&1 | grep -v amdgpu.ids
done; done
