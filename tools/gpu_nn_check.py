"""Neighbour search on large candidate sets: the store-once kernel against the multi-pass kernel (same library,
forced by size) and numpy, incl. low-dimensional / clustered / tied inputs; timing at n = 50000."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine
eng = Engine(0)
rng = np.random.default_rng(3)


def ref_query(q, x, m):
    d = ((q[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    idx = np.lexsort((np.broadcast_to(np.arange(x.shape[0]), d.shape), d), axis=1)   # (dist, index) order
    return idx[:, :m]


for name, x in (('uniform d=8', rng.uniform(size=(6000, 8))), ('uniform d=1', rng.uniform(size=(6000, 1))),
                ('clustered d=2', np.concatenate([rng.normal(size=(3000, 2)) * 0.01, rng.uniform(size=(3000, 2)) * 50])),
                ('grid with ties d=2', np.stack(np.meshgrid(np.arange(80.), np.arange(75.)), -1).reshape(-1, 2)),
                ('all equal', np.ones((5000, 3)))):
    q = x[rng.integers(0, len(x), 40)] + (0 if 'ties' in name or 'equal' in name else 1e-3)
    got = eng.nn_query(eng.tensor(q), eng.tensor(x), 30).cpu().numpy()
    # numpy distances are summed in a different order: compare as sets with the distance of the last neighbour as a check
    ref = ref_query(q, x, 30)
    same = np.mean([set(a) == set(b) for a, b in zip(got, ref)])
    dg = np.sort(((q[:, None, :] - x[got]) ** 2).sum(-1), 1)
    dr = np.sort(((q[:, None, :] - x[ref]) ** 2).sum(-1), 1)
    print('%-20s n=%d: identical neighbour sets %.2f, max |dist diff| %.1e' % (name, len(x), same, np.abs(dg - dr).max()))
    od = eng.nn_ordered(eng.tensor(x), 12).cpu().numpy()
    bad = 0
    for i in rng.integers(0, len(x), 60):
        d = ((x[:i + 1] - x[i]) ** 2).sum(1)
        want = np.sort(np.lexsort((np.arange(i + 1), d))[:13])[::-1]
        have = od[i][od[i] >= 0]
        dw, dh = np.sort(d[want]), np.sort(d[have])
        bad += not (len(want) == len(have) and np.allclose(dw, dh, rtol=0, atol=1e-12))
    print('   ordered NN rows checked: %d bad of 60' % bad)
n = 50000
X = eng.tensor(rng.uniform(size=(n, 8)))
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter(); NN = eng.nn_ordered(X, 25); torch.cuda.synchronize(); t1 = time.perf_counter() - t
    Q = eng.tensor(rng.uniform(size=(10000, 8)))
    torch.cuda.synchronize(); t = time.perf_counter(); PN = eng.nn_query(Q, X, 50); torch.cuda.synchronize(); t2 = time.perf_counter() - t
print('n=50000 d=8: ordered 25-NN %.3f s; 50 nearest of 50000 for 1e4 queries %.3f s' % (t1, t2))
