#!/bin/bash
# The Vecchia-path (BASELINE configs[3]: n = 50 000, d = 8, m = 25) measurements behind profiles/r02_cfg4_*.txt, on the GPU box:
#   bash tools/gpu_cfg4_artifacts.sh <out dir under gpurun_out>
# (rocprofv3 passes put the program itself after "--"; PMC passes are separate and carry no trace domains but --kernel-trace.)
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-cfg4}
mkdir -p "$O"
{
  echo "## python3 tools/gpu_vecchia_rowbench.py   (register-resident four-rows-per-wave kernels; HIP events, 20 launches each)"
  B=12 timeout 120 python3 tools/gpu_vecchia_rowbench.py 2>&1 | tail -2
  echo "## DGPAMD_VECCHIA_LDS=1 python3 tools/gpu_vecchia_rowbench.py   (the round-1 one-wave-per-row LDS kernels, same build)"
  DGPAMD_VECCHIA_LDS=1 B=12 timeout 120 python3 tools/gpu_vecchia_rowbench.py 2>&1 | tail -2
  echo "## python3 tools/gpu_nn_bench.py   (neighbour search: streaming top-k kernels / the store-once kernel)"
  timeout 120 python3 tools/gpu_nn_bench.py 2>&1 | tail -2
} > "$O/kernels.txt"
cat "$O/kernels.txt"
{
  echo "## TRAIN_ONLY=1 ITERS=40 python3 tools/gpu_scale_probe.py cfg4train"
  TRAIN_ONLY=1 ITERS=40 timeout 120 python3 tools/gpu_scale_probe.py cfg4train 2>&1 | grep cfg4train
  echo "## ITERS=12 python3 tools/gpu_scale_probe.py cfg4train"
  ITERS=12 timeout 120 python3 tools/gpu_scale_probe.py cfg4train 2>&1 | grep cfg4train
  echo "## python3 tools/gpu_cfg4_phases.py   (device synchronised around every phase: slower than the free-running loop)"
  timeout 120 python3 tools/gpu_cfg4_phases.py 2>&1 | tail -13
} > "$O/train.txt"
cat "$O/train.txt"
TRAIN_ONLY=1 ITERS=40 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg4prof -- python3 tools/gpu_scale_probe.py cfg4train > "$O/prof.log" 2>&1
{
  echo "## TRAIN_ONLY=1 ITERS=40 timeout 300 rocprofv3 --kernel-trace --stats -- python3 tools/gpu_scale_probe.py cfg4train ; tools/kernel_stats_top.py, tools/analyze_gaps.py (second half of the trace)"
  grep "cfg4train:" "$O/prof.log"
  python3 tools/kernel_stats_top.py /tmp/cfg4prof 14
  python3 tools/analyze_gaps.py /tmp/cfg4prof 50 | head -10
} > "$O/train_kernel_stats.txt"
cat "$O/train_kernel_stats.txt"
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  d=/tmp/pmc_$(echo $c | tr ' ' '_')
  B=12 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 tools/gpu_vecchia_rowbench.py > /dev/null 2>&1
done
{
  echo "## B=12 rocprofv3 --kernel-trace --pmc <counters, one pass each> -- python3 tools/gpu_vecchia_rowbench.py ; tools/pmc_kernel.py (per-launch averages)"
  for k in "vecchia_row4_kernel<0, 0, 26>" "vecchia_row4_kernel<0, 1, 26>" "vecchia_row4_kernel<1, 0, 26>"; do
    echo "== $k"
    python3 tools/pmc_kernel.py "$k" /tmp/pmc_*
  done
} > "$O/pmc_row_kernel.txt"
cat "$O/pmc_row_kernel.txt"
