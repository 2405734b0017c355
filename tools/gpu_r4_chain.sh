#!/bin/bash
# Round 4: the chain's second form against round 3's build, same box (run through gpurun from the repo root).
out=gpurun_out/r4_chain
mkdir -p $out
export PYTHONUNBUFFERED=1
R3=$PWD/build_ubench/libdgp_amd_r3.so
echo "== check (new build)"; timeout 600 python tools/gpu_mega_check.py > $out/check_new.txt 2>&1; echo rc=$?; tail -22 $out/check_new.txt
echo "== trace B=1"; timeout 120 python tools/gpu_mega_trace.py 2000 1 > $out/trace_b1.txt 2>&1; sed -n 1,12p $out/trace_b1.txt; sed -n 30,52p $out/trace_b1.txt
echo "== trace B=1 NOWAIT (timing only)"; DGPAMD_MEGA_NOWAIT=1 timeout 120 python tools/gpu_mega_trace.py 2000 1 > $out/trace_b1_nowait.txt 2>&1; sed -n 1,12p $out/trace_b1_nowait.txt; sed -n 34,37p $out/trace_b1_nowait.txt
echo "== trace B=1 inv"; timeout 120 python tools/gpu_mega_trace.py 2000 1 inv > $out/trace_b1_inv.txt 2>&1; sed -n 33,36p $out/trace_b1_inv.txt
echo "== trace B=12"; timeout 120 python tools/gpu_mega_trace.py 2000 12 > $out/trace_b12.txt 2>&1; sed -n 1,48p $out/trace_b12.txt
if [ -f $R3 ] && [ -z "$SKIP_R3" ]; then
  echo "== check (round 3 build)"; DGPAMD_LIB=$R3 timeout 600 python tools/gpu_mega_check.py > $out/check_r3.txt 2>&1; tail -8 $out/check_r3.txt
fi
echo "== LOOK=0 (new chain, look-ahead as three tasks)"; DGPAMD_MEGA_LOOK=0 timeout 600 python tools/gpu_mega_check.py > $out/check_look0.txt 2>&1; tail -8 $out/check_look0.txt
if [ -z "$SKIP_PYTEST" ]; then echo "== pytest potrf"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "potrf or workspace or mega or loglik" 2>&1 | tail -5; fi
