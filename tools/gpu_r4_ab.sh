#!/bin/bash
# A/B of two library builds on one box: build_ubench/libdgp_amd_base.so (before) against the in-tree library (after).
export PYTHONUNBUFFERED=1
BASE=$PWD/build_ubench/libdgp_amd_base.so
for lib in base new base new base new; do
  if [ $lib = base ]; then export DGPAMD_LIB=$BASE; else unset DGPAMD_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib bench steps20:', round(d['value'],3), 'frac', round(d['roofline']['frac'],4), 'inv B=1/6/12', [round(d['potrf_table'][k]['potrf_inv_ms'],3) for k in ('B=1','B=6','B=12')], 'rounds/iter', d['counts']['mstep_rounds_per_iter'])"
done
unset DGPAMD_LIB
if [ -z "$SKIP_CHECK" ]; then timeout 600 python tools/gpu_mega_check.py 2>&1 | tail -12; fi
