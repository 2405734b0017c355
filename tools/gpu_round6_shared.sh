#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out
{ echo "# round-6 build: (1) the linked-GP pair kernels under foreign load (tools/gpu_pair_shared.py: two other processes keep the GPU busy with factorisations and pair kernels; every result must equal the one computed alone bit for bit)"
  timeout 600 python tools/gpu_pair_shared.py 12 2>&1 | grep -v amdgpu.ids
  echo "# (2) the GPU suite in two processes side by side on one GPU (tools/gpu_suite_shared.sh)"
  timeout 2400 bash tools/gpu_suite_shared.sh 2>&1 | grep -v amdgpu.ids
  echo "# (3) three processes x 4000 launches of the one-launch factorisation sharing the GPU (tools/gpu_mega_stress_shared.sh)"
  timeout 900 bash tools/gpu_mega_stress_shared.sh 2>&1 | grep -v amdgpu.ids | tail -6
} > $O/r6_shared_gpu.txt 2>&1
cat $O/r6_shared_gpu.txt
