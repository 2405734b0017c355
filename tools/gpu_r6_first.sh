#!/bin/bash
# Round 6, first contact: the new parity tests, the bench line with step_split, a kernel trace of the training half for the idle-gap analysis.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -k "cfg2" tests/test_gpu_two_devices.py tests/test_gpu_bench_flow.py tests/test_isa_hazards.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -15 > $O/r6a_pytest.txt; cat $O/r6a_pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r6a_bench_steps20.json 2> $O/r6a_bench_steps20.err; tail -c 400 $O/r6a_bench_steps20.err
rm -rf /tmp/pf_trace; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernel none --sustained-steps 0 --no-predict > $O/r6a_trace_bench.json 2> $O/r6a_trace.err
f=$(find /tmp/pf_trace -name '*_kernel_trace.csv' | head -1)
python3 tools/analyze_gaps.py /tmp/pf_trace 30 > $O/r6a_idle_gaps.txt 2>&1
python3 tools/analyze_round.py /tmp/pf_trace >> $O/r6a_idle_gaps.txt 2>&1
python3 - "$f" $O/r6a_kernel_trace.tsv.gz <<'PY'
import csv, gzip, sys
with gzip.open(sys.argv[2], 'wt') as g:
    for r in csv.DictReader(open(sys.argv[1])):
        g.write('%s\t%s\t%s\n' % (r['Start_Timestamp'], r['End_Timestamp'], r['Kernel_Name'].split('(')[0][-48:]))
PY
head -2 "$f" > $O/r6a_kernel_trace_head.txt
ls -la $O | tail -12
