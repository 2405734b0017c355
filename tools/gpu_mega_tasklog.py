"""Full task log of the one-launch factorisation (dgpamd_debug_tasklog): every chain step and every worker task of every
matrix with its stamps, plus the task table the slots index, written to an .npz for tools/analyze_tasklog.py (which runs
anywhere).  usage: gpu_mega_tasklog.py out.npz n B [inv]   (environment: the DGPAMD_MEGA_* table overrides apply)"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine
from dgp_amd._lib import lib

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
inv = len(sys.argv) > 4 and sys.argv[4] == 'inv'
eng = Engine(0)
Np = eng.padded_dim(n)
nbk = Np // 64
rng = np.random.default_rng(0)
X = eng.tensor(rng.uniform(size=(B, n, 5)))
G = eng.tensor(rng.uniform(size=(n, 5)))
y = eng.tensor(rng.normal(size=n))
A = eng.empty(B, Np, Np)
work = eng.potrf_workspace(n, B)
T, S = (eng.empty(B, Np, Np), eng.empty(B, Np, Np)) if inv else (None, None)
eng.set_potrf_mode(1)
tab = np.zeros(8 * 40000 + 2 * nbk, dtype=np.int32)
ntask = lib.dgpamd_debug_mega_table(eng.h, n, int(inv), B, tab.ctypes.data_as(C.c_void_p), tab.size)
assert ntask > 0, ntask
words = 64 + 8 * (B * nbk + B * ntask)
log = torch.zeros(words, dtype=torch.int64, device=A.device)
ev0, ev1 = eng.event(), eng.event()
ms = []
for rep in range(4):
    eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
    if rep == 3:
        lib.dgpamd_debug_tasklog(eng.h, C.c_void_p(log.data_ptr()), words)
    eng.record(ev0)
    if inv:
        eng.potrf_inv(n, A, T, S, batch=B, work=work)
    else:
        eng.potrf(n, A, batch=B, work=work)
    eng.record(ev1)
    torch.cuda.synchronize()
    ms.append(eng.elapsed_ms(ev0, ev1))
lib.dgpamd_debug_tasklog(eng.h, None, 0)
raw = log.cpu().numpy()
np.savez_compressed(out, log=raw, table=tab[:8 * ntask].reshape(ntask, 8), need=tab[8 * ntask:8 * ntask + 2 * nbk].reshape(nbk, 2),
                    n=n, B=B, inv=int(inv), nbk=nbk, ntask=ntask, ms=np.array(ms))
print('n=%d B=%d inv=%s ntask=%d  ms (3 plain, 1 logged): %s' % (n, B, inv, ntask, ' '.join('%.3f' % v for v in ms)))
