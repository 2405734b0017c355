#!/bin/bash
# Everything profiles/r04_* is made from, on the GPU box: run from the repository root,
#   bash tools/gpu_round4_artifacts.sh <out dir under gpurun_out>
# (rocprofv3 passes put the program itself after "--"; PMC passes are separate and carry no trace domains but --kernel-trace.)
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-r04final}
mkdir -p "$O"
BENCH="bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernel none --min-gpu-seconds 0"
python3 bench.py --steps 20 --warmup 5 > "$O/bench_steps20.json" 2> "$O/bench_steps20.err"; tail -c 400 "$O/bench_steps20.json"; echo
python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; tail -c 300 "$O/bench_default.json"; echo
python3 tools/gpu_mega_check.py > "$O/potrf_modes.txt" 2>&1; tail -9 "$O/potrf_modes.txt"
{ python3 tools/gpu_mega_trace.py 2000 1; python3 tools/gpu_mega_trace.py 2000 6 inv; python3 tools/gpu_mega_trace.py 2000 12; } > "$O/potrf_phase_trace.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $BENCH > "$O/prof_stats.log" 2>&1
python3 tools/summarize_rocprof.py /tmp/prof_stats python3 $BENCH > "$O/bench_kernel_stats.txt" 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python3 $BENCH --no-predict > "$O/prof_train.log" 2>&1
echo "# rocprofv3 --kernel-trace -- python3 $BENCH --no-predict ; tools/analyze_gaps.py (last 70 % of the trace), analyze_round.py" > "$O/idle_gaps.txt"
python3 tools/analyze_gaps.py /tmp/prof_train 30 | head -16 >> "$O/idle_gaps.txt" 2>&1
python3 tools/analyze_round.py /tmp/prof_train >> "$O/idle_gaps.txt" 2>&1
head -6 "$O/idle_gaps.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_fetch --output-format csv -- python3 $BENCH > "$O/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_write --output-format csv -- python3 $BENCH > "$O/pmc_write.log" 2>&1
python3 tools/pmc_step_traffic.py potrf_mega_kernel /tmp/prof_fetch /tmp/prof_write "$O/pmc_bench_potrf_kernel.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 $BENCH" > /dev/null 2>&1
head -12 "$O/pmc_bench_potrf_kernel.json"
# ---- cfg4 (Vecchia, n = 50 000)
{
  echo "## ITER_TIMES=1 TRAIN_ONLY=1 ITERS=40 python3 tools/gpu_scale_probe.py cfg4train   (M-step started from the device state, two optimiser groups in flight)"
  ITER_TIMES=1 TRAIN_ONLY=1 ITERS=40 timeout 200 python3 tools/gpu_scale_probe.py cfg4train 2>&1 | grep cfg4train
  echo "## DGPAMD_MSTEP_EARLY=0 DGPAMD_MSTEP_GROUPS=1 DGPAMD_FETCH_SPIN=0 DGPAMD_PINNED_UPLOAD=0 ...   (round 3's host schedule)"
  DGPAMD_MSTEP_EARLY=0 DGPAMD_MSTEP_GROUPS=1 DGPAMD_FETCH_SPIN=0 DGPAMD_PINNED_UPLOAD=0 TRAIN_ONLY=1 ITERS=40 timeout 200 python3 tools/gpu_scale_probe.py cfg4train 2>&1 | grep cfg4train
  echo "## ITERS=12 MPRED=2000,20000,100000 python3 tools/gpu_scale_probe.py cfg4train   (prediction)"
  ITERS=12 MPRED=2000,20000,100000 timeout 300 python3 tools/gpu_scale_probe.py cfg4train 2>&1 | grep cfg4train
  echo "## python3 tools/gpu_cfg4_timeline.py   (host timeline of one steady-state iteration, no extra synchronisation)"
  MIN_US=150 timeout 120 python3 tools/gpu_cfg4_timeline.py 2>&1 | tail -40
} > "$O/cfg4_vecchia.txt"
cat "$O/cfg4_vecchia.txt"
EARLY=1 bash tools/gpu_r4_cfg4_trace.sh > /dev/null 2>&1; cp gpurun_out/r04/cfg4_train_kernel_stats_early1.txt "$O/cfg4_train_kernel_stats.txt"; head -24 "$O/cfg4_train_kernel_stats.txt"
timeout 1500 python3 -m pytest tests -q -m gpu > "$O/pytest_gpu.txt" 2>&1; tail -2 "$O/pytest_gpu.txt"
