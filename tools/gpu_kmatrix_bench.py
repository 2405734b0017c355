"""K-assembly throughput vs the HBM roofline across shapes (HIP events on the engine stream)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine

e = Engine(0)
rng = np.random.default_rng(0)
ev0, ev1 = e.event(), e.event()
print('%-10s %6s %3s %5s %3s | %9s %10s %8s' % ('kernel', 'n', 'D', 'mode', 'B', 'ms', 'GB/s(alg)', 'of 8TB/s'))
for name in ('sexp', 'matern2.5'):
    for (n, D, full, B) in ((2000, 5, False, 8), (2000, 10, False, 8), (2000, 10, True, 1), (5000, 10, True, 1), (5000, 10, False, 4),
                            (8192, 5, True, 1), (8192, 2, True, 1), (16384, 5, True, 1)):
        X = e.tensor(rng.uniform(size=(B, n, D)))
        ld = n if full else e.padded_dim(n)
        out = e.empty(B, ld, ld) if B > 1 else e.empty(ld, ld)
        length = np.full(D, 0.9)
        for _ in range(2):
            e.kmatrix(name, X if B > 1 else X[0], None, None, length, 1e-6, out=out, full=full, batch=B)
        reps = 10
        e.record(ev0)
        for _ in range(reps):
            e.kmatrix(name, X if B > 1 else X[0], None, None, length, 1e-6, out=out, full=full, batch=B)
        e.record(ev1)
        ms = e.elapsed_ms(ev0, ev1) / reps
        nbytes = B * ((8.0 * n * n) if full else (4.0 * ld * ld)) + 8.0 * B * n * D
        print('%-10s %6d %3d %5s %3d | %9.3f %10.0f %8.3f' % (name, n, D, 'full' if full else 'lower', B, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 8000))
        del out, X
