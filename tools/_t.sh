cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2p
python3 -m pytest tests -x -q -m gpu -k "mstep or lockstep or llik or train or iteration or grad" > gpurun_out/r2p/pytest_sel.txt 2>&1; tail -2 gpurun_out/r2p/pytest_sel.txt
python3 bench.py --no-cpu-baseline > gpurun_out/r2p/bench.json 2> gpurun_out/r2p/bench.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r2p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['predict']['pts_per_s'])
PY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict --prof-kernel none > gpurun_out/r2p/bench_prof.log 2>&1
python3 tools/analyze_round.py /tmp/prof > gpurun_out/r2p/rounds.txt 2>&1; cat gpurun_out/r2p/rounds.txt
python3 tools/analyze_gaps.py /tmp/prof 30 > gpurun_out/r2p/gaps.txt 2>&1; head -3 gpurun_out/r2p/gaps.txt
python3 tools/analyze_context.py /tmp/prof 120 > gpurun_out/r2p/ctx.txt 2>&1; head -5 gpurun_out/r2p/ctx.txt
