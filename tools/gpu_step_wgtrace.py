"""Per-workgroup timeline of ONE block-step launch of the factorisation (dgpamd_debug_trace): when each workgroup
started / ended, its task kind and panels, its CU.  usage: gpu_step_wgtrace.py n B k   (INV=1: fused inverse)"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine
from dgp_amd._lib import lib

eng = Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ks = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [4]
Np = eng.padded_dim(n)
rng = np.random.default_rng(0)
X = eng.tensor(rng.uniform(size=(B, n, 5)))
G = eng.tensor(rng.uniform(size=(n, 5)))
y = eng.tensor(rng.normal(size=n))
A = eng.empty(B, Np, Np)
work = eng.potrf_workspace(n, B)
inv = bool(os.environ.get('INV'))
if inv:
    T, S = eng.empty(B, Np, Np), eng.empty(B, Np, Np)
NW = 1 << 16
for kq in ks:
    tr = torch.zeros(4096 + 4 * NW, dtype=torch.int64, device=A.device)
    tr[4095] = kq
    for rep in range(3):
        eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
        if rep == 2:
            lib.dgpamd_debug_trace(eng.h, C.c_void_p(tr.data_ptr()))
            eng.set_graphs(False)
        if inv:
            eng.potrf_inv(n, A, T, S, batch=B, work=work)
        else:
            eng.potrf(n, A, batch=B, work=work)
    torch.cuda.synchronize()
    lib.dgpamd_debug_trace(eng.h, None)
    eng.set_graphs(True)
    w = tr.cpu().numpy()[4096:].reshape(-1, 4)
    w = w[w[:, 0] != 0]
    t0 = w[:, 0].min()
    st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
    kind, nkb, cu = w[:, 2] & 15, (w[:, 2] >> 8) & 255, w[:, 2] >> 16
    names = {0: 'store', 1: 'solve', 2: 'chain', 3: 'tdiag', 4: 'store2'}
    print('launch k=%d, B=%d%s: %d workgroups with a task, span %.1f us, %d distinct CUs' % (kq, B, ' (inverse)' if inv else '', len(w), en.max(), len(set(cu))))
    for kd in sorted(set(kind)):
        for nk in sorted(set(nkb[kind == kd])):
            m = (kind == kd) & (nkb == nk)
            d = en[m] - st[m]
            print('  %-5s nkb=%d: %5d tasks, duration mean %.1f us (min %.1f max %.1f), starts %.1f..%.1f, last end %.1f' % (
                names[int(kd)], nk, m.sum(), d.mean(), d.min(), d.max(), st[m].min(), st[m].max(), en[m].max()))
    # slot occupancy over time: number of running workgroups in 5-us bins
    edges = np.arange(0, en.max() + 5, 5.0)
    occ = [(np.minimum(en, b1) - np.maximum(st, b0)).clip(0).sum() / 5.0 for b0, b1 in zip(edges[:-1], edges[1:])]
    print('  running workgroups per 5-us bin:', ' '.join('%d' % round(o) for o in occ))
    busy = (en - st)[(kind == 0) | (kind == 4)].sum()
    print('  bulk workgroup-time %.0f us = %.1f us x 512 slots' % (busy, busy / 512))
