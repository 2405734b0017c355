"""In-kernel phase timestamps of one factorisation (dgpamd_debug_trace), matrix 0.  100 MHz wall clock -> us."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine
from dgp_amd._lib import lib

eng = Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Np = eng.padded_dim(n)
rng = np.random.default_rng(0)
X = eng.tensor(rng.uniform(size=(B, n, 5)))
G = eng.tensor(rng.uniform(size=(n, 5)))
y = eng.tensor(rng.normal(size=n))
A = eng.empty(B, Np, Np)
work = eng.potrf_workspace(n, B)
T, S = eng.empty(B, Np, Np), eng.empty(B, Np, Np)
tr = torch.zeros(4096, dtype=torch.int64, device=A.device)
for rep in range(3):
    eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
    if rep == 2:
        lib.dgpamd_debug_trace(eng.h, C.c_void_p(tr.data_ptr()))
        eng.set_graphs(False)
    if os.environ.get('INV'):
        eng.potrf_inv(n, A, T, S, batch=B, work=work)
    else:
        eng.potrf(n, A, batch=B, work=work)
torch.cuda.synchronize()
lib.dgpamd_debug_trace(eng.h, None)
t = tr.cpu().numpy().reshape(-1, 16).astype(np.float64) / 100.0   # us
nbk = Np // 64
print(' k | upd    diag   release | panel: wait  solve | start->next start')
for k in range(1, nbk):
    r = t[k]
    nxt = t[k + 1][0] - r[0] if k + 1 < nbk else float('nan')
    print('%2d | %5.1f  %5.1f  %5.1f | %5.1f %5.1f | %6.1f' % (k, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[9] - r[8] if r[8] else 0,
                                                        r[10] - r[9] if r[8] else 0, nxt))

d = tr.cpu().numpy()[1024:1024 + 16 * nbk].reshape(-1, 16).astype(np.float64) / 100.0
kk = 5
r = d[kk]
print('block %d, diagonal factor (us from its start): ' % kk + ' | '.join(
    '16-block %d: start %.2f, factored %.2f' % (J, r[2 * J] - r[0], r[2 * J + 1] - r[0]) for J in range(4)) + ' | end %.2f' % (r[8] - r[0]))
