export PYTHONUNBUFFERED=1
BASE=$PWD/build_ubench/libdgp_amd_base.so
for lib in base new base new; do
  if [ $lib = base ]; then export DGPAMD_LIB=$BASE; else unset DGPAMD_LIB; fi
  echo "== $lib"; B=6 python tools/gpu_vecchia_rowbench.py 2>&1 | tail -2; D=16 B=4 python tools/gpu_vecchia_rowbench.py 2>&1 | tail -2
  TRAIN_ONLY=1 ITERS=40 python tools/gpu_scale_probe.py cfg4train 2>&1 | tail -1
done
unset DGPAMD_LIB
python -m pytest tests -q -m gpu -k "vecch or golden or queue" 2>&1 | tail -3
