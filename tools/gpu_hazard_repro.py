"""Hunt for the round-2 hazard ('per-step graphs of two batch sizes replayed on one workspace after a one-launch
call: the first replay returned a wrong info'): one workspace shared by calls of different batch sizes and modes, results
checked without any synchronisation in between.  usage: gpu_hazard_repro.py [rounds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine

eng = Engine(0)
n = 2000
Np = eng.padded_dim(n)
rng = np.random.default_rng(0)
Bmax = 12
X = eng.tensor(rng.uniform(size=(Bmax, n, 5)))
y = eng.tensor(rng.normal(size=n))
A = eng.empty(Bmax, Np, Np)
work = eng.potrf_workspace(n, Bmax)       # ONE workspace for every batch size below
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
ref = {}
seq = [(1, 12), (0, 4), (0, 12), (1, 3), (0, 6), (2, 12), (2, 4), (0, 4), (1, 12), (0, 12)]
for r in range(rounds):
    outs = []
    for mode, B in seq:
        eng.set_potrf_mode(mode)
        eng.kmatrix('matern2.5', X[:B], None, None, [1.0], 1e-4, out=A[:B], full=False, Y=y, batch=B)
        ld, info = eng.potrf(n, A[:B], batch=B, work=work)
        outs.append((mode, B, ld, info))          # no synchronisation: the next call follows at once
    torch.cuda.synchronize()
    for mode, B, ld, info in outs:
        i, l = info.cpu().numpy(), ld.cpu().numpy()
        key = B
        if key not in ref:
            ref[key] = l.copy()
        if i.any() or not np.array_equal(l, ref[key]):
            bad += 1
            print('round %d: mode %d batch %d: info %s, logdet equal to the first run: %s' % (r, mode, B, i.tolist(), np.array_equal(l, ref[key])))
print('%d calls checked, %d wrong' % (rounds * len(seq), bad))
