"""Which tile goes wrong when several processes share the GPU?  Repeats one batched factorisation and compares every 64 x 64 tile
of L (and of K^-1 with INV=1) with the first run's; prints the first differing launches' tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgp_amd.ops import Engine
eng = Engine(0)
n, B, inv = int(os.environ.get('N', '150')), int(os.environ.get('B', '6')), os.environ.get('INV', '0') == '1'
r = np.random.default_rng(n * 100 + B)
X = eng.tensor(r.uniform(size=(B, n, 4))); G = eng.tensor(r.uniform(size=(n, 3))); y = eng.tensor(r.normal(size=n))
Np = eng.padded_dim(n); nb = Np // 64
A, T, S = eng.empty(B, Np, Np), eng.empty(B, Np, Np), eng.empty(B, Np, Np)
work = eng.potrf_workspace(n, B)
ref = None
shown = 0
for it in range(int(os.environ.get('REPS', '3000'))):
    eng.kmatrix('matern2.5', X, None, G, [0.7], 1e-5, out=A, full=False, Y=y, batch=B)
    if inv:
        ld, info = eng.potrf_inv(n, A, T, S, batch=B, work=work)
    else:
        ld, info = eng.potrf(n, A, batch=B, work=work)
    L = torch.tril(A[:, :n + 1, :n]).clone()
    if ref is None:
        ref = (L, ld.clone())
        continue
    if not torch.equal(L, ref[0]) or not torch.equal(ld, ref[1]):
        d = (L - ref[0]).abs()
        msg = []
        for b in range(B):
            for i in range(nb):
                for j in range(i + 1):
                    blk = d[b, 64 * i:64 * i + 64, 64 * j:64 * j + 64]
                    if blk.numel() and float(blk.max()) > 0:
                        rows = torch.nonzero(blk.max(1).values > 0).flatten().tolist()
                        cols = torch.nonzero(blk.max(0).values > 0).flatten().tolist()
                        msg.append('matrix %d tile (%d,%d): max |d| %.2e, rows %d..%d (%d), cols %d..%d (%d)' % (b, i, j, float(blk.max()), rows[0], rows[-1], len(rows), cols[0], cols[-1], len(cols)))
        print('launch %d differs: logdet d %s info %s\n   ' % (it, (ld - ref[1]).cpu().numpy(), info.cpu().numpy()) + '\n   '.join(msg[:8]), flush=True)
        shown += 1
        if shown >= 4:
            break
print('pid %d done: %d differing launches shown' % (os.getpid(), shown))
