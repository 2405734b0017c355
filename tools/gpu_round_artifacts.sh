#!/bin/bash
# Everything the round's profiles/ files are made from, on the GPU box: run from the repository root,
#   bash tools/gpu_round_artifacts.sh <out dir under gpurun_out>
# (rocprofv3 passes put the program itself after "--"; PMC passes are separate and carry no trace domains but --kernel-trace.)
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-final}
mkdir -p "$O"
BENCH="bench.py --steps 20 --warmup 3 --no-cpu-baseline --prof-kernel none"
python3 -m pytest tests -x -q -m gpu > "$O/pytest_gpu.txt" 2>&1; tail -2 "$O/pytest_gpu.txt"
python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; tail -c 600 "$O/bench_default.json"; echo
python3 tools/gpu_mega_check.py > "$O/potrf_modes.txt" 2>&1; tail -9 "$O/potrf_modes.txt"
python3 tools/gpu_mega_trace.py 2000 1 > "$O/phase_trace_b1.txt" 2>&1
python3 tools/gpu_mega_trace.py 2000 6 inv > "$O/phase_trace_b6inv.txt" 2>&1
python3 tools/gpu_mega_trace.py 2000 12 > "$O/phase_trace_b12.txt" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $BENCH > "$O/prof_stats.log" 2>&1
python3 tools/summarize_rocprof.py /tmp/prof_stats python3 $BENCH > "$O/bench_kernel_stats.txt" 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python3 $BENCH --no-predict > "$O/prof_train.log" 2>&1
echo "# rocprofv3 --kernel-trace -- python3 $BENCH --no-predict ; tools/analyze_gaps.py (last 70 % of the trace), analyze_round.py, analyze_context.py" > "$O/idle_gaps.txt"
python3 tools/analyze_gaps.py /tmp/prof_train 30 >> "$O/idle_gaps.txt" 2>&1
python3 tools/analyze_round.py /tmp/prof_train >> "$O/idle_gaps.txt" 2>&1
python3 tools/analyze_context.py /tmp/prof_train 120 >> "$O/idle_gaps.txt" 2>&1
head -4 "$O/idle_gaps.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_fetch --output-format csv -- python3 $BENCH > "$O/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_write --output-format csv -- python3 $BENCH > "$O/pmc_write.log" 2>&1
python3 tools/pmc_step_traffic.py potrf_mega_kernel /tmp/prof_fetch /tmp/prof_write "$O/pmc_bench_potrf_kernel.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 $BENCH" > /dev/null 2>&1
cat "$O/pmc_bench_potrf_kernel.json" | head -12
