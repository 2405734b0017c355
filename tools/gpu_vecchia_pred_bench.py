"""Vecchia prediction kernels alone at cfg4's shape (n = 50 000 training points, pm = 50 neighbours, 100 000 test points):
gp_vecch (D = 8) and link_gp_vecch (Dw = Dz = 8, squared exponential), register-resident kernels vs the LDS kernels
(DGPAMD_VECCHIA_LDS=1), HIP-event timed; results compared."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine
eng = Engine(0)
rng = np.random.default_rng(3)
n, M, pm = 50000, int(os.environ.get('M', 100000)), 50
X = rng.uniform(size=(n, 8)); W = rng.normal(size=(n, 8)); y = rng.normal(size=n)
xq = rng.uniform(size=(M, 8)); mm = rng.normal(size=(M, 8)); vv = rng.uniform(0.01, 0.3, size=(M, 8))
ones = eng.tensor(np.ones(n))
dX, dW, dy, dq, dm, dv = (eng.tensor(a) for a in (X, W, y, xq, mm, vv))
NN1 = eng.nn_query(dq, dX, pm)
NN2 = eng.nn_query(eng.tensor(np.concatenate((mm, xq), 1)), eng.tensor(np.concatenate((W, X), 1)), pm)
ev0, ev1 = eng.event(), eng.event()


def timed(f):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        eng.record(ev0); out = f(); eng.record(ev1); torch.cuda.synchronize(); ts.append(eng.elapsed_ms(ev0, ev1))
    return min(ts), out


res = {}
kind = os.environ.get('KIND', 'sexp')
for lds in ('0', '1'):
    os.environ['DGPAMD_VECCHIA_LDS'] = lds
    tg, og = timed(lambda: eng.vecchia_gp(kind, dq, dX, NN1, dy, 1.3, np.ones(1), 1e-4, ones))
    tl, ol = timed(lambda: eng.vecchia_linkgp(kind, dm, dv, dq, dW, dX, NN2, dy, 1.3, np.array([2.0]), 1e-4, ones))
    res[lds] = (tg, tl, [t.cpu().numpy() for t in og + ol])
    print('%s kernels: gp_vecch %.2f ms, link_gp_vecch %.2f ms per %d points (pm = %d)' % ('LDS     ' if lds == '1' else 'register', tg, tl, M, pm))
d = [float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300))) for a, b in zip(res['0'][2], res['1'][2])]
print('max rel diff register vs LDS: gp mean %.1e var %.1e, link_gp mean %.1e var %.1e' % tuple(d))
