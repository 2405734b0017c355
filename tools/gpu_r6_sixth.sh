#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out
for b in default 10,6,2 10,6,3 8,4,4; do
  if [ $b = default ]; then unset DGPAMD_ESS_BATCH; else export DGPAMD_ESS_BATCH=$b; fi
  python tools/gpu_chain_fingerprint.py 10 2>&1 | grep -v amdgpu.ids
done > $O/r6f_chain_fingerprint.txt
cat $O/r6f_chain_fingerprint.txt
