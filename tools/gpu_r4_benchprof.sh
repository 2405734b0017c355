cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r04; mkdir -p $O
BENCH="bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernel none --min-gpu-seconds 0 --no-predict"
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-predict 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench steps20:', d['value'], d['ms_per_step'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python3 $BENCH > "$O/prof_train.log" 2>&1
python3 tools/kernel_stats_top.py /tmp/prof_train 16
python3 tools/analyze_gaps.py /tmp/prof_train 30 | head -14
python3 tools/analyze_round.py /tmp/prof_train 2>&1 | head -30
