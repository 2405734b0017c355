"""potrf / potri wall time (HIP events, min of 5) against the batch size at the bench size."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine

eng = Engine(0)
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
Np = eng.padded_dim(n)
G = eng.tensor(rng.uniform(size=(n, 5)))
y = eng.tensor(rng.normal(size=n))
ev0, ev1 = eng.event(), eng.event()
print('n=%d   B | potrf ms  TF/s | potri ms  TF/s | potrf_inv (one sweep) ms  TF/s' % n)
for B in (1, 2, 3, 4, 6, 8, 12):
    X = eng.tensor(rng.uniform(size=(B, n, 5)))
    A = eng.empty(B, Np, Np)
    Ainv = eng.empty(B, Np, Np)
    work = eng.potrf_workspace(n, B)
    T = eng.empty(B, Np, Np)
    tf, ti, tv = [], [], []
    for rep in range(6):
        eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
        eng.record(ev0)
        eng.potrf(n, A, batch=B, work=work)
        eng.record(ev1)
        tf.append(eng.elapsed_ms(ev0, ev1))
        eng.record(ev0)
        eng.potri(n, A, Ainv, 1, work, batch=B)
        eng.record(ev1)
        ti.append(eng.elapsed_ms(ev0, ev1))
        eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
        eng.record(ev0)
        eng.potrf_inv(n, A, T, Ainv, batch=B, work=work)
        eng.record(ev1)
        tv.append(eng.elapsed_ms(ev0, ev1))
    print('        %2d | %7.3f %6.2f | %7.3f %6.2f | %7.3f %6.2f' % (B, min(tf), B * n ** 3 / 3 / min(tf) / 1e9, min(ti), B * 2 * n ** 3 / 3 / min(ti) / 1e9,
                                                             min(tv), B * n ** 3 / min(tv) / 1e9))
