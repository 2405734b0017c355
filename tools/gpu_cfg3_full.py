"""BASELINE configs[2] at its full size on ONE GPU: 2-layer DGP, d = 10 in / 3 out, n = 5000 (default structure: SExp),
S imputations (50), M test points (1e5).  Prints training speed, emulator construction, prediction time and peak memory.
usage: S=50 M=100000 python tools/gpu_cfg3_full.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd import dgp, emulator

n, d, q = int(os.environ.get('N', '5000')), 10, 3
S, M = int(os.environ.get('S', '50')), int(os.environ.get('M', '100000'))
rng = np.random.default_rng(2026)
X = rng.uniform(size=(n, d))
Y = np.stack([np.sin(1.0 / ((0.7 * X[:, 0] + 0.3) * (0.7 * X[:, 1] + 0.3))) + (0.2 + 0.1 * j) * (X[:, 2 + j:] ** 2).sum(1) for j in range(q)], 1)
Y = (Y - Y.mean(0)) / Y.std(0)
sync = torch.cuda.synchronize
t = time.perf_counter(); model = dgp(X, Y, seed=1); sync()
print('cfg3 n=%d: construct (warm start + 11 sweeps) %.2f s' % (n, time.perf_counter() - t), flush=True)
its = int(os.environ.get('ITERS', '4'))
model.train(N=1, ess_burn=10, disable=True); sync()
t = time.perf_counter(); model.train(N=its, ess_burn=10, disable=True); sync(); dt = time.perf_counter() - t
print('cfg3: %d SI iterations %.2f s -> %.2f it/s' % (its, dt, its / dt), flush=True)
t = time.perf_counter(); emu = emulator(model.estimate(burnin=0), N=S, seed=3); sync()
print('cfg3: emulator(N=%d) %.1f s' % (S, time.perf_counter() - t), flush=True)
xt = rng.uniform(size=(M, d))
t = time.perf_counter(); emu.predict(xt[:64]); sync()
print('cfg3: statistics + first predict %.1f s, memory %.1f GB' % (time.perf_counter() - t, torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
t = time.perf_counter(); mu, var = emu.predict(xt); sync(); dt = time.perf_counter() - t
print('cfg3: predict %d points x %d imputations: %.1f s -> %.0f pts/s, %.0f point-imputations/s; finite %s; peak memory %.1f GB'
      % (M, S, dt, M / dt, M * S / dt, bool(np.all(np.isfinite(mu)) and np.all(np.isfinite(var))), torch.cuda.max_memory_allocated() / 2 ** 30), flush=True)
