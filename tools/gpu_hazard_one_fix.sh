#!/bin/bash
# Round 4's two fixes of the chain, one at a time: the shared-GPU stress (three processes, every result
# compared with the first run of the same inputs) against the shipped build, a build with round 4's first store form only
# (-DMEGA_DIAG_OLD_STORE=1: one set of data registers, offsets in SGPRs; second-look fix kept) and a build with the first
# second-look form only (-DMEGA_DIAG_OLD_LOOK=1: the bits written into sh.ready; store fix kept).
# The diagnostic libraries are built by hand from csrc/chol.hip with those macros into build_ubench/ (never shipped).
cd "$GRAFT_REPO_ROOT"
for lib in "" build_ubench/libdgp_amd_diag_OLD_STORE.so build_ubench/libdgp_amd_diag_OLD_LOOK.so; do
  echo "#### library: ${lib:-dgp_amd/libdgp_amd.so (shipped)}"
  if [ -n "$lib" ]; then export DGPAMD_LIB="$GRAFT_REPO_ROOT/$lib"; else unset DGPAMD_LIB; fi
  LAUNCHES=${LAUNCHES:-4000} bash tools/gpu_mega_stress_shared.sh "X=1"
done
