#!/bin/bash
# Where do the cycles of the Matern pair kernel go?  One rocprofv3 --pmc pass per counter set over tools/gpu_linkgp_bench.py at the
# bench's shape (n = 2000, 5 inputs, cell order, 4096 points = 16 launches + warm-up).   bash tools/gpu_pair_pmc.sh <out dir under gpurun_out> [PIPE]
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp ORDER=1 CHECK=0 DGPAMD_JSEP_PIPE=${2:-0}
O=gpurun_out/${1:-r5pairpmc}
mkdir -p "$O"
CMD="tools/gpu_linkgp_bench.py matern2.5 2000 5 0 4096"
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_CYCLES")
i=0
for c in "${SETS[@]}"; do
  rm -rf /tmp/pp_$i; rocprofv3 --kernel-trace --pmc $c -d /tmp/pp_$i --output-format csv -- python3 $CMD > "$O/pass$i.log" 2>&1
  i=$((i+1))
done
{ echo "# rocprofv3 --kernel-trace --pmc <set> -- python3 $CMD   (ORDER=1 CHECK=0 DGPAMD_JSEP_PIPE=$DGPAMD_JSEP_PIPE; one pass per set; per-launch averages)"
  j=0; while [ $j -lt $i ]; do python3 tools/pmc_kernel.py linkgp_Jsep_kernel /tmp/pp_$j; j=$((j+1)); done; } > "$O/pmc_pair_kernel.txt" 2>&1
cat "$O/pmc_pair_kernel.txt"
