#!/bin/bash
# Everything profiles/r03_* is made from, on the GPU box: run from the repository root,
#   bash tools/gpu_round3_artifacts.sh <out dir under gpurun_out>
# (rocprofv3 passes put the program itself after "--"; PMC passes are separate and carry no trace domains but --kernel-trace.)
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-r03final}
mkdir -p "$O"
BENCH="bench.py --steps 20 --warmup 3 --no-cpu-baseline --prof-kernel none --min-gpu-seconds 0"
python3 bench.py --steps 20 --warmup 5 > "$O/bench_steps20.json" 2> "$O/bench_steps20.err"; tail -c 400 "$O/bench_steps20.json"; echo
python3 bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; tail -c 300 "$O/bench_default.json"; echo
python3 tools/gpu_mega_check.py > "$O/potrf_modes.txt" 2>&1; tail -9 "$O/potrf_modes.txt"
{ python3 tools/gpu_mega_trace.py 2000 1; python3 tools/gpu_mega_trace.py 2000 6 inv; python3 tools/gpu_mega_trace.py 2000 12; } > "$O/potrf_phase_trace.txt" 2>&1
python3 tools/gpu_mstep_nodes.py > "$O/mstep_nodes.txt" 2>&1; tail -8 "$O/mstep_nodes.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $BENCH > "$O/prof_stats.log" 2>&1
python3 tools/summarize_rocprof.py /tmp/prof_stats python3 $BENCH > "$O/bench_kernel_stats.txt" 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_train -- python3 $BENCH --no-predict > "$O/prof_train.log" 2>&1
echo "# rocprofv3 --kernel-trace -- python3 $BENCH --no-predict ; tools/analyze_gaps.py (last 70 % of the trace), analyze_round.py, analyze_context.py" > "$O/idle_gaps.txt"
python3 tools/analyze_gaps.py /tmp/prof_train 30 >> "$O/idle_gaps.txt" 2>&1
python3 tools/analyze_round.py /tmp/prof_train >> "$O/idle_gaps.txt" 2>&1
head -6 "$O/idle_gaps.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_fetch --output-format csv -- python3 $BENCH > "$O/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_write --output-format csv -- python3 $BENCH > "$O/pmc_write.log" 2>&1
python3 tools/pmc_step_traffic.py potrf_mega_kernel /tmp/prof_fetch /tmp/prof_write "$O/pmc_bench_potrf_kernel.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 $BENCH" > /dev/null 2>&1
head -12 "$O/pmc_bench_potrf_kernel.json"
# ---- cfg4 (Vecchia, n = 50 000)
{
  echo "## TRAIN_ONLY=1 ITERS=40 python3 tools/gpu_scale_probe.py cfg4train   (device-resident ESS queue, level-scheduled draws)"
  TRAIN_ONLY=1 ITERS=40 timeout 200 python3 tools/gpu_scale_probe.py cfg4train 2>&1 | grep cfg4train
  echo "## NOQUEUE=1 TRAIN_ONLY=1 ITERS=40 python3 tools/gpu_scale_probe.py cfg4train   (host-driven accept / shrink loop, round 2's path)"
  NOQUEUE=1 TRAIN_ONLY=1 ITERS=40 timeout 200 python3 tools/gpu_scale_probe.py cfg4train 2>&1 | grep cfg4train
  echo "## ITERS=12 MPRED=2000,20000,100000 python3 tools/gpu_scale_probe.py cfg4train   (prediction: shared neighbour searches)"
  ITERS=12 MPRED=2000,20000,100000 timeout 300 python3 tools/gpu_scale_probe.py cfg4train 2>&1 | grep cfg4train
  echo "## DGPAMD_NN_SHARE=0 ITERS=12 MPRED=100000 ...   (one search per node and imputation, as round 2)"
  DGPAMD_NN_SHARE=0 ITERS=12 MPRED=100000 timeout 300 python3 tools/gpu_scale_probe.py cfg4train 2>&1 | grep predict
  echo "## python3 tools/gpu_spsolve_bench.py"
  timeout 120 python3 tools/gpu_spsolve_bench.py 2>&1 | tail -1
  echo "## python3 tools/gpu_cfg4_phases.py   (device synchronised around every phase)"
  timeout 120 python3 tools/gpu_cfg4_phases.py 2>&1 | tail -14
} > "$O/cfg4_vecchia.txt"
cat "$O/cfg4_vecchia.txt"
{
  echo "## python3 tools/gpu_vecchia_pred_bench.py"
  timeout 300 python3 tools/gpu_vecchia_pred_bench.py 2>&1 | tail -3
  echo "## ITERS=3 MPRED=100000 rocprofv3 --kernel-trace --stats -- python3 tools/gpu_scale_probe.py cfg4train ; tools/kernel_stats_top.py (training of 3 iterations + prediction)"
  ITERS=3 MPRED=100000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg4pred -- python3 tools/gpu_scale_probe.py cfg4train > /tmp/cfg4pred.log 2>&1
  grep predict /tmp/cfg4pred.log
  python3 tools/kernel_stats_top.py /tmp/cfg4pred 12
} > "$O/cfg4_predict_kernels.txt" 2>&1
cat "$O/cfg4_predict_kernels.txt"
TRAIN_ONLY=1 ITERS=40 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg4prof -- python3 tools/gpu_scale_probe.py cfg4train > "$O/cfg4_prof.log" 2>&1
{
  echo "## TRAIN_ONLY=1 ITERS=40 rocprofv3 --kernel-trace --stats -- python3 tools/gpu_scale_probe.py cfg4train ; tools/kernel_stats_top.py, tools/analyze_gaps.py (second half of the trace)"
  grep cfg4train "$O/cfg4_prof.log"
  python3 tools/kernel_stats_top.py /tmp/cfg4prof 14
  python3 tools/analyze_gaps.py /tmp/cfg4prof 50 | head -14
} > "$O/cfg4_train_kernel_stats.txt" 2>&1
head -8 "$O/cfg4_train_kernel_stats.txt"
# ---- linked-GP pair kernels
{
  echo "## python3 tools/gpu_linkgp_bench.py sexp 5000 10 10 2048   (second form: records + MFMA-absorbed row / column terms)"
  timeout 200 python3 tools/gpu_linkgp_bench.py sexp 5000 10 10 2048 2>&1 | tail -2
  echo "## DGPAMD_SEXP_FORM1=1 ...   (round 2's kernel with the full-rate exp)"
  DGPAMD_SEXP_FORM1=1 timeout 200 python3 tools/gpu_linkgp_bench.py sexp 5000 10 10 2048 2>&1 | tail -2
  echo "## python3 tools/gpu_linkgp_bench.py matern2.5 2000 5 5 4096   (training points in the caller's order: every step mixed)"
  CHECK=0 timeout 200 python3 tools/gpu_linkgp_bench.py matern2.5 2000 5 5 4096 2>&1 | tail -1
  echo "## ORDER=1 ...   (training points grouped by cells, as the emulator hands them over: order classes)"
  ORDER=1 CHECK=0 timeout 200 python3 tools/gpu_linkgp_bench.py matern2.5 2000 5 5 4096 2>&1 | tail -1
  echo "## python3 tools/gpu_kmatrix_bench.py"
  timeout 200 python3 tools/gpu_kmatrix_bench.py 2>&1 | tail -17
} > "$O/pair_kernels.txt"
cat "$O/pair_kernels.txt"
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do rm -rf /tmp/pm; CHECK=0 rocprofv3 --kernel-trace --pmc $c -d /tmp/pm --output-format csv -- python3 tools/gpu_linkgp_bench.py sexp 5000 10 10 2048 > /dev/null 2>&1; python3 tools/pmc_kernel.py linkgp_Jsexp2 /tmp/pm; done > "$O/pmc_sexp_pair_kernel.txt" 2>&1
cat "$O/pmc_sexp_pair_kernel.txt"
timeout 1500 python3 -m pytest tests -q -m gpu > "$O/pytest_gpu.txt" 2>&1; tail -2 "$O/pytest_gpu.txt"
