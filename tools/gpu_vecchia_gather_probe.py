"""Is the Vecchia row kernel bound by its gather?  The same point set (a 2-D sheet embedded in D dimensions) ordered along a Morton
curve (a row's neighbours sit at nearby indices: the gather hits L1 / L2) and in random order (what the reference's
np.random.permutation ordering gives: every neighbour row is its own cache line somewhere in the array)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import default_engine
eng = default_engine(0)
n, m, B = 50000, 25, int(os.environ.get('B', '4'))
rng = np.random.default_rng(1)
uv = rng.uniform(size=(n, 2))


def morton(u, v, bits=10):
    a, b = (u * (1 << bits)).astype(np.int64), (v * (1 << bits)).astype(np.int64)
    code = np.zeros(len(u), dtype=np.int64)
    for i in range(bits):
        code |= ((a >> i) & 1) << (2 * i) | ((b >> i) & 1) << (2 * i + 1)
    return code


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


for D in (8, 16):
    emb = np.stack([np.sin((k + 1) * uv[:, 0] + 0.3 * k) * np.cos((k % 3 + 1) * uv[:, 1]) for k in range(D)], 1) + 0.01 * rng.normal(size=(n, D))
    for name, order in (('Morton order', np.argsort(morton(uv[:, 0], uv[:, 1]))), ('random order', rng.permutation(n))):
        X = emb[order]
        y = np.sin(3 * uv[order, 0]) + 0.1 * rng.normal(size=n)
        length = np.array([0.5])
        NN = eng.nn_ordered(eng.tensor(X / length), m)
        dX, dy, ones = eng.tensor(X), eng.tensor(y), eng.tensor(np.ones(n))
        XB = dX.unsqueeze(0).repeat(B, 1, 1).contiguous()
        span = float((torch.arange(n, device=NN.device)[:, None] - NN).clamp(min=0).double().mean())
        t1 = timed(lambda: eng.vecchia_llik('sexp', dX, dy, NN, length, 1e-4, ones))
        tb = timed(lambda: eng.vecchia_llik_batch('sexp', XB, dy, NN, length, 1e-4, ones))
        t2 = timed(lambda: eng.vecchia_nllik('sexp', dX, dy, NN, length, 1e-4, ones, True))
        print('D=%2d %s (mean index distance to a neighbour %7.0f): llik %.0f us | llik x%d %.0f us | nllik %.0f us' % (D, name, span, t1, B, tb, t2))
