"""Time the factorisation pipeline pieces at the bench size (run on the GPU box)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine

eng = Engine(0)
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
Np = eng.padded_dim(n)
for B in (1, 8):
    X = eng.tensor(rng.uniform(size=(B, n, 5)))
    G = eng.tensor(rng.uniform(size=(n, 5)))
    y = eng.tensor(rng.normal(size=n))
    A = eng.empty(B, Np, Np)
    for name in ('kmatrix', 'syrk'):
        for rep in range(3):
            eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
            eng.prof_enable(name)
            if name == 'kmatrix':
                eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
            else:
                eng.potrf(n, A, batch=B)
            k, ms, w = eng.prof_collect()
        print('n=%d B=%d %-10s launches %3d total %.3f ms avg %.1f us  %.2f %s' % (
            n, B, name, k, ms, 1e3 * ms / k, w / ms / (1e6 if name == 'kmatrix' else 1e9), 'GB/s' if name == 'kmatrix' else 'TFLOP/s'))
    ev0, ev1 = eng.event(), eng.event()
    ts = []
    for rep in range(5):
        eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
        eng.record(ev0)
        ld, info = eng.potrf(n, A, batch=B)
        eng.record(ev1)
        ts.append(eng.elapsed_ms(ev0, ev1))
    print('n=%d B=%d potrf total %.3f ms (min of 5)  = %.2f TFLOP/s algorithmic (n^3/3 per matrix)' % (n, B, min(ts), B * n ** 3 / 3 / min(ts) / 1e9))
    # inverse
    if B == 1:
        work = eng.potrf_workspace(n, 1)
        Ainv = eng.empty(Np, Np)
        ts = []
        for rep in range(3):
            eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=1)
            eng.potrf(n, A, batch=1, work=work)
            eng.record(ev0)
            eng.potri(n, A[0], Ainv, 1, work)
            eng.record(ev1)
            ts.append(eng.elapsed_ms(ev0, ev1))
        print('n=%d potri %.3f ms' % (n, min(ts)))
