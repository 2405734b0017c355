#!/bin/bash
# sweep of the one-launch factorisation's table parameters: one process per setting (the table is cached per process)
# usage: gpu_r5_sweep.sh "VAR=a,b VAR2=c ..." BLIST   -> every combination listed on the command line as KEY=V1,V2
out=${OUT:-gpurun_out/r5_sweep.txt}
BL=${BLIST:-1,2,3,4}
for setting in "$@"; do
  echo "== $setting" >> $out
  env $setting BLIST=$BL timeout 300 python tools/gpu_lazy_sweep.py run 2>&1 | grep -v amdgpu >> $out
done
cat $out
