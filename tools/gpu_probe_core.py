"""First-light probe of the core HIP path against numpy/scipy (run on the GPU box)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dgp_oracle as O

lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dgp_amd', 'libdgp_amd.so'))
lib.dgpamd_padded_dim.restype = C.c_int64
lib.dgpamd_padded_dim.argtypes = [C.c_int64]
lib.dgpamd_potrf_workspace.restype = C.c_size_t
lib.dgpamd_potrf_workspace.argtypes = [C.c_int64, C.c_int]
lib.dgpamd_grad_workspace.restype = C.c_size_t
lib.dgpamd_grad_workspace.argtypes = [C.c_int64, C.c_int]
lib.dgpamd_last_error.restype = C.c_char_p
p, i, l, d = C.c_void_p, C.c_int, C.c_int64, C.c_double
lib.dgpamd_kmatrix.argtypes = [p, i, l, p, l, l, p, i, p, i, p, i, d, p, p, l, l, i, p, l, l, i, i]
lib.dgpamd_potrf.argtypes = [p, l, p, l, i, p, p, p]
lib.dgpamd_potri.argtypes = [p, l, p, p, i, p]
lib.dgpamd_loglik.argtypes = [p, i, l, p, l, l, p, i, p, i, p, i, d, p, d, p, p, l, i, p, p, p]
lib.dgpamd_grad_reduce.argtypes = [p, i, l, p, l, p, i, p, i, p, i, d, p, i, p, p, p]
lib.dgpamd_trmv_lower.argtypes = [p, l, p, l, p, p, p, i]
lib.dgpamd_create.argtypes = [i, p, C.POINTER(p)]
lib.dgpamd_sync.argtypes = [p]
lib.dgpamd_last_error.argtypes = [p]

ctx = p()
assert lib.dgpamd_create(0, None, C.byref(ctx)) == 0
dev = torch.device('cuda:0')


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def chk(rc):
    if rc != 0:
        raise RuntimeError('rc=%d %s' % (rc, lib.dgpamd_last_error(ctx).decode()))


def hptr(a):
    return a.ctypes.data_as(C.c_void_p)


rng = np.random.default_rng(0)
for kind, name in ((0, 'sexp'), (1, 'matern2.5')):
    for n, Dl, Dg, per_dim in ((50, 2, 0, False), (200, 3, 2, True), (333, 5, 5, False), (1024, 2, 1, True)):
        D = Dl + Dg
        Xl = rng.uniform(size=(n, Dl + 1))     # one unused column: exercises colmap
        cm = np.arange(1, Dl + 1, dtype=np.int32)
        Xg = rng.uniform(size=(n, Dg)) if Dg else None
        X = Xl[:, cm] if Xg is None else np.concatenate((Xl[:, cm], Xg), 1)
        length = rng.uniform(0.5, 1.5, size=D if per_dim else 1)
        nugget = 1e-3
        y = rng.normal(size=n)
        Kref = O.k_matrix(X, length, nugget, name)
        # full K
        dXl, dXg = T(Xl), (T(Xg) if Dg else None)
        Kd = torch.empty((n, n), dtype=torch.float64, device=dev)
        chk(lib.dgpamd_kmatrix(ctx, kind, n, dXl.data_ptr(), Dl + 1, 0, hptr(cm), Dl, dXg.data_ptr() if Dg else None, Dg,
                               hptr(length), len(length), nugget, None, Kd.data_ptr(), n, 0, 1, None, 0, 0, 0, 1))
        lib.dgpamd_sync(ctx)
        err_k = np.abs(Kd.cpu().numpy() - Kref).max()
        # augmented + potrf
        Np = lib.dgpamd_padded_dim(n)
        A = torch.full((Np, Np), float('nan'), dtype=torch.float64, device=dev)
        dy = T(y)
        chk(lib.dgpamd_kmatrix(ctx, kind, n, dXl.data_ptr(), Dl + 1, 0, hptr(cm), Dl, dXg.data_ptr() if Dg else None, Dg,
                               hptr(length), len(length), nugget, None, A.data_ptr(), Np, 0, 0, dy.data_ptr(), n, 0, 1, 1))
        ws = torch.empty(lib.dgpamd_potrf_workspace(n, 1), dtype=torch.uint8, device=dev)
        logdet = torch.zeros(1, dtype=torch.float64, device=dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        chk(lib.dgpamd_potrf(ctx, n, A.data_ptr(), Np * Np, 1, logdet.data_ptr(), info.data_ptr(), ws.data_ptr()))
        lib.dgpamd_sync(ctx)
        Lref = np.linalg.cholesky(Kref)
        Ah = A.cpu().numpy()
        err_l = np.abs(np.tril(Ah[:n, :n]) - Lref).max()
        wref = np.linalg.solve(Lref, y)
        err_w = np.abs(Ah[n, :n] - wref).max()
        err_q = abs(-Ah[n, n] - wref @ wref) / (wref @ wref)
        err_ld = abs(logdet.item() - 2 * np.log(np.diag(Lref)).sum())
        # trmv
        z = rng.normal(size=n)
        out = torch.empty(n, dtype=torch.float64, device=dev)
        sc = np.array([1.7])
        chk(lib.dgpamd_trmv_lower(ctx, n, A.data_ptr(), Np * Np, hptr(sc), T(z).data_ptr(), out.data_ptr(), 1))
        lib.dgpamd_sync(ctx)
        err_mv = np.abs(out.cpu().numpy() - np.sqrt(1.7) * Lref @ z).max()
        # inverse
        Ainv = torch.full((Np, Np), float('nan'), dtype=torch.float64, device=dev)
        chk(lib.dgpamd_potri(ctx, n, A.data_ptr(), Ainv.data_ptr(), 1, ws.data_ptr()))
        lib.dgpamd_sync(ctx)
        Kinv_ref = np.linalg.inv(Kref)
        Aih = Ainv.cpu().numpy()
        err_inv = np.abs(Aih[:n, :n] - Kinv_ref).max() / np.abs(Kinv_ref).max()
        alpha_ref = Kinv_ref @ y
        err_al = np.abs(-Aih[n, :n] - alpha_ref).max() / np.abs(alpha_ref).max()
        # gradient reductions
        for nugget_est in (0, 1):
            P = (1 if not per_dim else D) + nugget_est
            gws = torch.empty(lib.dgpamd_grad_workspace(n, P), dtype=torch.uint8, device=dev)
            gout = torch.empty(2 * P, dtype=torch.float64, device=dev)
            chk(lib.dgpamd_grad_reduce(ctx, kind, n, dXl.data_ptr(), Dl + 1, hptr(cm), Dl, dXg.data_ptr() if Dg else None, Dg,
                                       hptr(length), len(length), nugget, None, nugget_est, Ainv.data_ptr(), gout.data_ptr(),
                                       gws.data_ptr()))
            lib.dgpamd_sync(ctx)
            _, fod = O.k_matrix_fod(X, length, nugget, name, bool(nugget_est))
            tr_ref = np.array([np.sum(Kinv_ref * f) for f in fod])
            q_ref = np.array([alpha_ref @ f @ alpha_ref for f in fod])
            g = gout.cpu().numpy()
            err_tr = np.abs(g[:P] - tr_ref).max() / (np.abs(tr_ref).max() + 1e-300)
            err_qd = np.abs(g[P:] - q_ref).max() / (np.abs(q_ref).max() + 1e-300)
            print('   grad nugget_est=%d  tr %.2e  quad %.2e' % (nugget_est, err_tr, err_qd))
        # batched loglik
        B = 3
        A3 = torch.empty((B, Np, Np), dtype=torch.float64, device=dev)
        Xb = np.stack([Xl + 0.01 * b for b in range(B)])
        ws3 = torch.empty(lib.dgpamd_potrf_workspace(n, B), dtype=torch.uint8, device=dev)
        ll = torch.empty(B, dtype=torch.float64, device=dev)
        info3 = torch.zeros(B, dtype=torch.int32, device=dev)
        dXb = T(Xb)
        chk(lib.dgpamd_loglik(ctx, kind, n, dXb.data_ptr(), Dl + 1, n * (Dl + 1), hptr(cm), Dl, dXg.data_ptr() if Dg else None, Dg,
                              hptr(length), len(length), nugget, None, 1.7, dy.data_ptr(), A3.data_ptr(), Np * Np, B,
                              ll.data_ptr(), info3.data_ptr(), ws3.data_ptr()))
        lib.dgpamd_sync(ctx)
        llr = []
        for b in range(B):
            Xbb = Xb[b][:, cm] if Xg is None else np.concatenate((Xb[b][:, cm], Xg), 1)
            llr.append(O.log_likelihood(Xbb, y, length, 1.7, nugget, name))
        err_ll = np.abs(ll.cpu().numpy() - np.array(llr)).max() / np.abs(llr).max()
        print('%s n=%d D=%d per_dim=%d: K %.1e L %.1e w %.1e quad %.1e logdet %.1e trmv %.1e inv %.1e alpha %.1e ll %.1e info %s'
              % (name, n, D, per_dim, err_k, err_l, err_w, err_q, err_ld, err_mv, err_inv, err_al, err_ll, info3.cpu().numpy()))

# timing at the bench size
n, B = 2000, 8
Np = lib.dgpamd_padded_dim(n)
Xl = rng.uniform(size=(B, n, 5))
Xg = rng.uniform(size=(n, 5))
y = rng.normal(size=n)
A3 = torch.empty((B, Np, Np), dtype=torch.float64, device=dev)
ws3 = torch.empty(lib.dgpamd_potrf_workspace(n, B), dtype=torch.uint8, device=dev)
ll = torch.empty(B, dtype=torch.float64, device=dev)
info3 = torch.zeros(B, dtype=torch.int32, device=dev)
dXb, dXg, dy = T(Xl), T(Xg), T(y)
length = np.array([1.0])
for nb in (1, 8):
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.time()
        chk(lib.dgpamd_loglik(ctx, 1, n, dXb.data_ptr(), 5, n * 5, None, 5, dXg.data_ptr(), 5, hptr(length), 1, 1e-6, None, 1.0,
                              dy.data_ptr(), A3.data_ptr(), Np * Np, nb, ll.data_ptr(), info3.data_ptr(), ws3.data_ptr()))
        lib.dgpamd_sync(ctx)
        print('loglik n=2000 batch=%d: %.3f ms' % (nb, (time.time() - t) * 1e3), ll.cpu().numpy()[:2], info3.cpu().numpy()[:nb])
t0 = time.time()
refll = O.log_likelihood(np.concatenate((Xl[0], Xg), 1), y, length, 1.0, 1e-6, 'matern2.5')
print('cpu oracle loglik n=2000: %.3f s value %.6f' % (time.time() - t0, refll))
