"""Where a wave of the Vecchia row kernel spends its time: shader-clock stamps (s_memtime) of the first 64 row blocks of one
vecchia_llik launch (dgpamd_debug_trace): neighbour row + gather + staging | barrier | pair loop | barrier | block into registers |
factorisation."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import default_engine
from dgp_amd._lib import lib
eng = default_engine(0)
n, m = 50000, 25
rng = np.random.default_rng(7)
for name in ('sexp', 'matern2.5'):
    for d in (8, 16):
        X = rng.uniform(size=(n, d)); y = np.sin(3 * X[:, 0]) + 0.1 * rng.normal(size=n); length = np.array([0.8])
        NN = eng.nn_ordered(eng.tensor(X / length), m)
        dX, dy, ones = eng.tensor(X), eng.tensor(y), eng.tensor(np.ones(n))
        for mode in ('llik', 'nllik'):
            f = (lambda: eng.vecchia_llik(name, dX, dy, NN, length, 1e-4, ones)) if mode == 'llik' else (lambda: eng.vecchia_nllik(name, dX, dy, NN, length, 1e-4, ones, True))
            f(); torch.cuda.synchronize()
            tr = torch.zeros(8192, dtype=torch.int64, device=eng.device)
            lib.dgpamd_debug_trace(eng.h, C.c_void_p(tr.data_ptr()))
            f(); torch.cuda.synchronize()
            lib.dgpamd_debug_trace(eng.h, None)
            full = tr.cpu().numpy()[:64 * 16].reshape(64, 16)[8:].astype(float)   # (row blocks 8..63: full conditioning sets)
            st = full[:, :7]
            dt = np.diff(st, axis=1)
            med = np.median(dt, axis=0)
            tail = '' if mode == 'llik' else ' | back-substitutions %6.0f | derivative sums %6.0f' % (np.median(full[:, 7] - full[:, 6]), np.median(full[:, 8] - full[:, 7]))
            print('%-9s d=%2d %-5s cycles (median of 56 waves): gather+stage %6.0f | barrier %5.0f | pair loop %6.0f | barrier %5.0f | to registers %5.0f | factorisation %6.0f%s | total %6.0f'
                  % (name, d, mode, med[0], med[1], med[2], med[3], med[4], med[5], tail, np.median(full[:, 8 if mode == 'nllik' else 6] - full[:, 0])))
