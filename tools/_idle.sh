cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2idle
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict --prof-kernel none > gpurun_out/r2idle/bench.log 2>&1
python3 tools/analyze_gaps.py /tmp/prof 30 > gpurun_out/r2idle/gaps.txt 2>&1
python3 tools/analyze_round.py /tmp/prof > gpurun_out/r2idle/rounds.txt 2>&1
cat gpurun_out/r2idle/rounds.txt
