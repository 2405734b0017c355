#!/bin/bash
# A/B of two library builds on one box: build_ubench/libdgp_amd_base.so (before) against the in-tree library (after).
export PYTHONUNBUFFERED=1
BASE=$PWD/build_ubench/libdgp_amd_base.so
for lib in base new base new; do
  if [ $lib = base ]; then export DGPAMD_LIB=$BASE; else unset DGPAMD_LIB; fi
  echo "== $lib"
  timeout 120 python tools/gpu_mega_trace.py 2000 1 2>&1 | sed -n 30,40p
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench steps20:', round(d['value'],3), d['potrf_table']['B=1'], d['potrf_table']['B=6'])"
done
unset DGPAMD_LIB
timeout 600 python tools/gpu_mega_check.py 2>&1 | tail -12
