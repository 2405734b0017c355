#!/bin/bash
# Round 6, second run: the rest of the new tests, K assembly with the table exponential against the library exp (same box), the wave-priority experiment.
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py::test_cfg2_emulator_predict_of_the_bench_model_vs_oracle_walk tests/test_gpu_two_devices.py tests/test_gpu_bench_flow.py tests/test_gpu_ops.py tests/test_gpu_shims.py -m gpu -q 2>&1 | tail -25 > $O/r6b_pytest.txt; cat $O/r6b_pytest.txt
export SHAPES="5000,10,0,10,2 2000,5,0,12,3 2000,10,0,6,3 5000,10,1,1,3 8192,10,1,1,3 16384,10,1,1,2"
echo "# table exponential (this build)" > $O/r6b_kmatrix_ab.txt
timeout 600 python tools/gpu_kmatrix_roofline.py >> $O/r6b_kmatrix_ab.txt 2>&1
echo "# library exp (-DKM_EXP_TAB=0: rounds 1-5), same box" >> $O/r6b_kmatrix_ab.txt
DGPAMD_LIB=$PWD/build_ubench/libdgp_amd_exp0.so timeout 600 python tools/gpu_kmatrix_roofline.py >> $O/r6b_kmatrix_ab.txt 2>&1
echo "# table exponential again" >> $O/r6b_kmatrix_ab.txt
timeout 600 python tools/gpu_kmatrix_roofline.py >> $O/r6b_kmatrix_ab.txt 2>&1
cat $O/r6b_kmatrix_ab.txt
timeout 900 python tools/gpu_prio_ab.py 2000 9 > $O/r6b_prio_ab.txt 2>&1; cat $O/r6b_prio_ab.txt
for p in 0 1 3; do
  DGPAMD_MEGA_PRIO=$p timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prof-kernel none --sustained-steps 0 --no-predict > $O/r6b_bench_prio$p.json 2> $O/r6b_bench_prio$p.err
  python3 -c "import json;d=json.load(open('$O/r6b_bench_prio$p.json'));print('prio $p', d['value'], d['ms_per_step'], d['step_split']['parts_ms_per_step']['istep_ms'], d['counts']['mstep_rounds_per_iter'])"
done
