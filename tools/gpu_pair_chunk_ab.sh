for rep in 1 2; do
for c in 256 0; do
  echo "## DGPAMD_PAIR_CHUNK=$c"
  if [ $c = 0 ]; then unset DGPAMD_PAIR_CHUNK; else export DGPAMD_PAIR_CHUNK=$c; fi
  ORDER=1 CHECK=$([ $rep = 1 ] && echo 1 || echo 0) python tools/gpu_linkgp_bench.py matern2.5 2000 5 0 4096 2>&1 | grep -v amdgpu.ids
  CHECK=0 python tools/gpu_linkgp_bench.py sexp 5000 10 10 2048 2>&1 | grep -v amdgpu.ids
  CHECK=0 ORDER=1 python tools/gpu_linkgp_bench.py matern2.5 5000 10 0 1024 2>&1 | grep -v amdgpu.ids
done; done
