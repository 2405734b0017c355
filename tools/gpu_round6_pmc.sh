#!/bin/bash
# Round 6 counter evidence (as round 5's, tools/gpu_round5_pmc.sh, plus the training form of K assembly and the SExp pair kernel): rocprofv3 --pmc passes (the program itself after "--", one counter set per pass, only
# --kernel-trace beside --pmc) for the four kernels the verdict names, plus the FETCH_SIZE calibration per load width.
#   bash tools/gpu_round6_pmc.sh <out dir under gpurun_out>
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-r6pmc}
mkdir -p "$O"
BENCH="bench.py --steps 10 --warmup 3 --no-cpu-baseline --prof-kernel none --sustained-steps 0 --predict-points 4096 --predict-seconds 0.3"
SETS=("SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE")
# (1) calibration of FETCH_SIZE per load width
rm -rf /tmp/pm_cal; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pm_cal --output-format csv -- build_ubench/fetch_calib > "$O/calib.log" 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE -- build_ubench/fetch_calib   (every kernel streams 1 GiB = 1048576 KB once per launch; FETCH_SIZE is in KB)"
  for k in calib_b128_plain calib_b128_sc1 calib_b64_sc1; do echo "== $k"; python3 tools/pmc_kernel.py $k /tmp/pm_cal; done; } > "$O/fetch_calibration.txt" 2>&1
cat "$O/fetch_calibration.txt"
# (2) the bench (training + a short prediction leg): potrf_mega_kernel, kmatrix_multi_kernel / kmatrix_kernel, grad_reduce, linkgp_Jsep_kernel
i=0
for c in "${SETS[@]}"; do
  rm -rf /tmp/pm_b$i; rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_b$i --output-format csv -- python3 $BENCH > "$O/bench_pass$i.log" 2>&1
  i=$((i+1))
done
{ echo "# rocprofv3 --kernel-trace --pmc <set> -- python3 $BENCH   (one pass per set: ${SETS[*]}; tools/pmc_kernel.py: per-launch averages over ALL launches of the kernel in the run)"
  for k in potrf_mega_kernel kmatrix_multi_kernel kmatrix_kernel grad_reduce linkgp_Jsep_kernel matern_records_kernel gp_quad_kernel; do
    echo "== $k"; for j in 0 1 2 3 4 5; do python3 tools/pmc_kernel.py $k /tmp/pm_b$j; done
  done; } > "$O/pmc_bench_kernels.txt" 2>&1
# kernel durations of the same command (for the per-launch times the counters are divided by)
rm -rf /tmp/pm_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_stats -- python3 $BENCH > "$O/bench_stats.log" 2>&1
python3 tools/summarize_rocprof.py /tmp/pm_stats python3 $BENCH > "$O/bench_kernel_stats.txt" 2>&1
# (3) the Vecchia row kernel at cfg4's shape
i=0
for c in "${SETS[@]}"; do
  rm -rf /tmp/pm_v$i; rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_v$i --output-format csv -- python3 tools/gpu_vecchia_rowbench.py > "$O/vecchia_pass$i.log" 2>&1
  i=$((i+1))
done
{ echo "# rocprofv3 --kernel-trace --pmc <set> -- python3 tools/gpu_vecchia_rowbench.py   (n = 50 000, d = 8, m = 25)"
  for k in vecchia_row4_kernel; do echo "== $k"; for j in 0 1 2 3 4 5; do python3 tools/pmc_kernel.py $k /tmp/pm_v$j; done; done; } > "$O/pmc_vecchia_kernels.txt" 2>&1
rm -rf /tmp/pm_vs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_vs -- python3 tools/gpu_vecchia_rowbench.py > "$O/vecchia_stats.log" 2>&1
python3 tools/summarize_rocprof.py /tmp/pm_vs python3 tools/gpu_vecchia_rowbench.py > "$O/vecchia_kernel_stats.txt" 2>&1
# (4) the stand-alone K assembly (full symmetric n = 8192, D = 10)
i=0
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "WRITE_SIZE" "FETCH_SIZE"; do
  rm -rf /tmp/pm_k$i; SHAPES="8192,10,1,1,3" rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_k$i --output-format csv -- python3 tools/gpu_kmatrix_roofline.py > "$O/kmat_pass$i.log" 2>&1
  i=$((i+1))
done
{ echo "# SHAPES=8192,10,1,1,3 rocprofv3 --kernel-trace --pmc <set> -- python3 tools/gpu_kmatrix_roofline.py   (full symmetric n = 8192, D = 10: 537 MB per launch)"
  for k in "kmatrix_kernel<0>" "kmatrix_kernel<1>"; do echo "== $k"; for j in 0 1 2 3; do python3 tools/pmc_kernel.py "$k" /tmp/pm_k$j; done; done; } > "$O/pmc_kmatrix_standalone.txt" 2>&1
# (5) K assembly in the form the training path uses: lower tiles, ten matrices, n = 5000, D = 10
i=0
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "WRITE_SIZE" "FETCH_SIZE"; do
  rm -rf /tmp/pm_l$i; SHAPES="5000,10,0,10,2" rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_l$i --output-format csv -- python3 tools/gpu_kmatrix_roofline.py > "$O/kmatl_pass$i.log" 2>&1
  i=$((i+1))
done
{ echo "# SHAPES=5000,10,0,10,2 rocprofv3 --kernel-trace --pmc <set> -- python3 tools/gpu_kmatrix_roofline.py   (lower tiles of ten n = 5000 matrices, D = 10: 1.05 GB per launch)"
  for k in "kmatrix_kernel<0>" "kmatrix_kernel<1>"; do echo "== $k"; for j in 0 1 2 3; do python3 tools/pmc_kernel.py "$k" /tmp/pm_l$j; done; done; } > "$O/pmc_kmatrix_lower.txt" 2>&1
# (6) the SExp pair kernel at cfg3's second-layer shape
i=0
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pm_s$i; CHECK=0 rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_s$i --output-format csv -- python3 tools/gpu_linkgp_bench.py sexp 5000 10 10 2048 > "$O/sexp_pass$i.log" 2>&1
  i=$((i+1))
done
{ echo "# CHECK=0 rocprofv3 --kernel-trace --pmc <set> -- python3 tools/gpu_linkgp_bench.py sexp 5000 10 10 2048   (cfg3's second layer: n = 5000, 10 + 10 inputs, 2048 test points)"
  for k in "linkgp_Jsexp2_kernel"; do echo "== $k"; for j in 0 1 2 3 4; do python3 tools/pmc_kernel.py "$k" /tmp/pm_s$j; done; done; } > "$O/pmc_sexp_pair.txt" 2>&1
tail -5 "$O/pmc_bench_kernels.txt" "$O/pmc_vecchia_kernels.txt" "$O/pmc_kmatrix_standalone.txt" "$O/pmc_kmatrix_lower.txt" "$O/pmc_sexp_pair.txt"
