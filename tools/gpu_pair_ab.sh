#!/bin/bash
# Same-box A/B of the Matern pair kernel's variants (DGPAMD_JSEP_PIPE) at the bench's shape (n = 2000, 5 local + 0 global inputs,
# training points in cell order) and at a larger one.   usage: gpu_pair_ab.sh out.txt "0 1 2"
out=${1:-gpurun_out/r5_pair_ab.txt}; vars=${2:-"0 1 2"}
: > $out
for rep in 1 2; do
  for v in $vars; do
    echo "## PIPE=$v rep $rep" >> $out
    ORDER=1 CHECK=$([ $rep = 1 ] && echo 1 || echo 0) DGPAMD_JSEP_PIPE=$v python tools/gpu_linkgp_bench.py matern2.5 2000 5 0 4096 2>&1 | grep -v amdgpu.ids >> $out
    ORDER=1 CHECK=0 DGPAMD_JSEP_PIPE=$v python tools/gpu_linkgp_bench.py matern2.5 5000 10 2 512 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
