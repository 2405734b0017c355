#!/bin/bash
# cfg4 training (n = 50 000 Vecchia, 40 SI iterations) under rocprofv3 --kernel-trace: kernel table + idle-gap classes of the
# second half of the trace.  Output: gpurun_out/r04/cfg4_train_kernel_stats.txt
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
for e in ${EARLY:-1}; do
rm -rf /tmp/cfg4prof
DGPAMD_MSTEP_EARLY=$e TRAIN_ONLY=1 ITERS=40 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg4prof -- python3 tools/gpu_scale_probe.py cfg4train > "$O/cfg4_prof.log" 2>&1
{
  echo "## DGPAMD_MSTEP_EARLY=$e TRAIN_ONLY=1 ITERS=40 rocprofv3 --kernel-trace --stats -- python3 tools/gpu_scale_probe.py cfg4train ; tools/kernel_stats_top.py, tools/analyze_gaps.py (second half of the trace)"
  grep cfg4train "$O/cfg4_prof.log"
  python3 tools/kernel_stats_top.py /tmp/cfg4prof 14
  python3 tools/analyze_gaps.py /tmp/cfg4prof 50 | head -16; python3 tools/analyze_gaps.py /tmp/cfg4prof 70 1500 4 | sed -n "/^---/,\$p"
} > "$O/cfg4_train_kernel_stats_early$e.txt" 2>&1
cat "$O/cfg4_train_kernel_stats_early$e.txt"
done
