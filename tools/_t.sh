cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2o
python3 -m pytest tests -x -q -m gpu > gpurun_out/r2o/pytest.txt 2>&1; tail -3 gpurun_out/r2o/pytest.txt
python3 bench.py --no-cpu-baseline > gpurun_out/r2o/bench.json 2> gpurun_out/r2o/bench.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r2o/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['predict']['pts_per_s'])
PY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-predict --prof-kernel none > gpurun_out/r2o/bench_prof.log 2>&1
python3 tools/analyze_gaps.py /tmp/prof 30 > gpurun_out/r2o/gaps.txt 2>&1
head -12 gpurun_out/r2o/gaps.txt
