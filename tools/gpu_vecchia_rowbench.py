"""Per-kernel time of the Vecchia row computations at BASELINE configs[3]'s shape (n = 50 000, d = 8, m = 25), HIP-event timed."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import default_engine

eng = default_engine(0)
n, d, m = int(os.environ.get('N', '50000')), int(os.environ.get('D', '8')), int(os.environ.get('M', '25'))
B = int(os.environ.get('B', '4'))
rng = np.random.default_rng(7)
X = rng.uniform(size=(n, d))
y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 + 0.1 * rng.normal(size=n)
length = np.array([0.8])
NN = eng.nn_ordered(eng.tensor(X / length), m)
dX, dy, ones = eng.tensor(X), eng.tensor(y), eng.tensor(np.ones(n))
XB = dX.unsqueeze(0).repeat(B, 1, 1).contiguous()


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    with eng.stream():
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            fn()
        e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name in ('sexp', 'matern2.5'):
    t1 = timed(lambda: eng.vecchia_llik(name, dX, dy, NN, length, 1e-4, ones))
    tb = timed(lambda: eng.vecchia_llik_batch(name, XB, dy, NN, length, 1e-4, ones))
    t2 = timed(lambda: eng.vecchia_nllik(name, dX, dy, NN, length, 1e-4, ones, True))
    t3 = timed(lambda: eng.vecchia_lmatrix(name, dX, NN, length, 1e-4))
    print('%s n=%d d=%d m=%d: llik %.0f us (%.1f Mrows/s) | llik x%d %.0f us (%.1f Mrows/s) | nllik(P=2) %.0f us | L_matrix %.0f us'
          % (name, n, d, m, t1, n / t1, B, tb, B * n / tb, t2, t3))
