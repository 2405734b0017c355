"""Sparse forward substitution at the cfg4 shape (n = 50 000, m = 25; 8 nodes x 11 sweeps as one I-step's prior draws):
row-by-row windows against the level schedule."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine
eng = Engine(0)
n, d, m = int(os.environ.get('N', '50000')), 8, 25
rng = np.random.default_rng(1)
X = rng.uniform(size=(n, d))
NN = eng.nn_ordered(eng.tensor(X), m)
Lm = eng.vecchia_lmatrix('sexp', eng.tensor(X), NN, np.array([1.0]), 1e-6)
nmat, nrhs = 8, 11
Ls, NNs = torch.stack([Lm] * nmat), torch.stack([NN] * nmat)
b = eng.tensor(rng.normal(size=(nmat, nrhs, n)))
sc = [1.0] * nmat
torch.cuda.synchronize(); t = time.perf_counter()
sched = eng.vecchia_levels(NNs)
torch.cuda.synchronize(); t_lev = time.perf_counter() - t
nlev = int(sched[4 * n + 2].item())


def bench(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps, r


t0, x0 = bench(lambda: eng.vecchia_spsolve_batch(Ls, NNs, sc, b))
t1, x1 = bench(lambda: eng.vecchia_spsolve_levels(Ls, NNs, sc, b, sched))
err = float((x0 - x1).abs().max() / x0.abs().max())
print('n=%d m=%d, %d matrices x %d right-hand sides: row windows %.2f ms, level schedule %.2f ms (%d levels; building the schedules %.1f ms); '
      'max rel diff %.1e' % (n, m, nmat, nrhs, 1e3 * t0, 1e3 * t1, nlev, 1e3 * t_lev, err))
