#!/bin/bash
# Round-5 final evidence on one box: the GPU suite, the driver's bench command, the default bench command, kernel stats of the driver's command
# (rocprofv3 --kernel-trace --stats), then the counter passes (tools/gpu_round5_pmc.sh).   bash tools/gpu_round5_final.sh
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/r5_pytest_gpu.txt; cat $O/r5_pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/r5_bench_steps20.json 2> $O/r5_bench_steps20.err; tail -c 600 $O/r5_bench_steps20.err
python bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err
rm -rf /tmp/pf_stats; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernel none --sustained-steps 0 > $O/r5_bench_stats.log 2>&1
python3 tools/summarize_rocprof.py /tmp/pf_stats python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernel none --sustained-steps 0 > $O/r5_bench_kernel_stats.txt 2>&1
head -12 $O/r5_bench_kernel_stats.txt
bash tools/gpu_round5_pmc.sh r5pmc2 > $O/r5pmc2.log 2>&1
tail -3 $O/r5pmc2.log
