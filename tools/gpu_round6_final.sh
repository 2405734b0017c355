#!/bin/bash
# Round-6 final evidence on one box: the GPU suite, the driver's bench command, the 20-step bench, a kernel trace of the training half (idle gaps, step account),
# kernel stats of the driver-style command, the train(N=500) soak, cfg3 at its stated size, then the counter passes (tools/gpu_round6_pmc.sh).
#   bash tools/gpu_round6_final.sh
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -24 > $O/r6_pytest_gpu.txt; tail -3 $O/r6_pytest_gpu.txt
timeout 900 python bench.py > $O/r6_bench_default.json 2> $O/r6_bench_default.err; tail -c 300 $O/r6_bench_default.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r6_bench_steps20.json 2> $O/r6_bench_steps20.err
python3 -c "import json;d=json.load(open('$O/r6_bench_default.json'));print('default: it/s', round(d['value'],2), 'sustained', round(d['sustained_it_per_s'],2), 'pts/s', round(d['predict']['pts_per_s'],1), 'frac', round(d['roofline']['frac'],4), 'cpu', d['cpu_baseline']['value'], d['cpu_baseline_predict'].get('value'))"
TR="bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernel none --sustained-steps 0 --no-predict"
rm -rf /tmp/pf_trace; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_trace -- python3 $TR > $O/r6_trace_bench.json 2> $O/r6_trace.err
{ echo "# rocprofv3 --kernel-trace -- python3 $TR ; tools/analyze_step.py (the iterations behind the first 25 % of the trace), tools/analyze_gaps.py, tools/analyze_round.py"
  echo "# the same run's own line (host clock, no profiler hooks in the timed region's arithmetic -- but the profiler's launch interception is in it):"
  python3 -c "import json;d=json.load(open('$O/r6_trace_bench.json'));s=d['step_split'];print('#   value', round(d['value'],2), 'SI it/s, ms_per_step', round(d['ms_per_step'],3), {k:round(v,3) for k,v in s['parts_ms_per_step'].items()}, 'gpu timeline', {k:round(v,3) for k,v in s['gpu_timeline_ms_per_step'].items()}, 'rounds', d['counts']['mstep_rounds_per_iter'])"
  python3 tools/analyze_step.py /tmp/pf_trace 25
  echo; echo "# idle time by (previous kernel -> next kernel), same trace:"
  python3 tools/analyze_gaps.py /tmp/pf_trace 25
  python3 tools/analyze_round.py /tmp/pf_trace
} > $O/r6_idle_gaps.txt 2>&1
head -30 $O/r6_idle_gaps.txt
ST="bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernel none --sustained-steps 0"
rm -rf /tmp/pf_stats; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -- python3 $ST > $O/r6_bench_stats.log 2>&1
python3 tools/summarize_rocprof.py /tmp/pf_stats python3 $ST > $O/r6_bench_kernel_stats.txt 2>&1
head -8 $O/r6_bench_kernel_stats.txt
timeout 900 python tools/gpu_soak.py 500 > $O/r6_train500_soak.txt 2>&1; tail -4 $O/r6_train500_soak.txt
S=50 M=100000 timeout 1200 python tools/gpu_cfg3_full.py 2>&1 | grep -v amdgpu.ids > $O/r6_cfg3_full_scale.txt; cat $O/r6_cfg3_full_scale.txt
bash tools/gpu_round6_pmc.sh r6pmc > $O/r6pmc.log 2>&1
tail -3 $O/r6pmc.log
python3 tools/pmc_summary.py $O/r6pmc $O/r6_pmc_kernels.json r06 > $O/r6_pmc_kernels.txt 2> $O/r6_pmc_summary.err; tail -3 $O/r6_pmc_summary.err; head -20 $O/r6_pmc_kernels.txt
