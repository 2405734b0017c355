#!/bin/bash
# The one-launch factorisation with several processes sharing the GPU (what the two-rank tests do): small matrices, every result
# compared bit for bit with the first run of the same inputs.  VARIANTS: environment settings to compare.
cd "$GRAFT_REPO_ROOT"
run() {
  for p in 1 2 3; do
    env $1 SIZES=${SIZES:-100,150,190,200,260,333} SEED=$((10 + p)) LAUNCHES=${LAUNCHES:-4000} python tools/gpu_mega_stress.py > /tmp/stress_$p.txt 2>&1 &
  done
  wait
  bad=0
  for p in 1 2 3; do grep -q "all results consistent" /tmp/stress_$p.txt || { bad=$((bad+1)); grep -m1 "AssertionError" /tmp/stress_$p.txt | cut -c1-160; }; done
  echo "## $1: $bad of 3 processes saw a differing result"
}
for v in "${@:-X=1}"; do run "$v"; run "$v"; done
