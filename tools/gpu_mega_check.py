"""One-launch factorisation (potrf mode 1) against the per-step launches (mode 0) and LAPACK; run-to-run bitwise
determinism (a stale tile read would show up as a difference); timings."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine

eng = Engine(0)
rng = np.random.default_rng(0)
ok = True


def build(n, B, seed=0):
    r = np.random.default_rng(seed)
    X = eng.tensor(r.uniform(size=(B, n, 5)))
    G = eng.tensor(r.uniform(size=(n, 5)))
    y = eng.tensor(r.normal(size=n))
    return X, G, y


for n, B in ((64, 1), (130, 2), (200, 1), (1000, 3), (2000, 1), (2000, 6), (2000, 12), (1999, 2), (2047, 1), (2048, 1)):
    Np = eng.padded_dim(n)
    X, G, y = build(n, B, n)
    work = eng.potrf_workspace(n, B)
    res = {}
    for mode in (0, 1):
        eng.set_potrf_mode(mode)
        A = eng.empty(B, Np, Np)
        eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-4, out=A, full=False, Y=y, batch=B)
        ld, info = eng.potrf(n, A, batch=B, work=work)
        torch.cuda.synchronize()
        T, S = eng.empty(B, Np, Np), eng.empty(B, Np, Np)
        T.fill_(float('nan')); S.fill_(float('nan'))   # the results must not depend on what these buffers held
        A2 = eng.empty(B, Np, Np)
        eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-4, out=A2, full=False, Y=y, batch=B)
        ld2, info2 = eng.potrf_inv(n, A2, T, S, batch=B, work=work)
        torch.cuda.synchronize()
        res[mode] = (A.cpu().numpy(), ld.cpu().numpy(), info.cpu().numpy(), S.cpu().numpy(), ld2.cpu().numpy(), info2.cpu().numpy(), A2.cpu().numpy())
    for mode in (0, 1):
        Sm = res[mode][3]
        nn = [int(np.isnan(np.tril(Sm[b][:n, :n])).sum()) + int(np.isnan(Sm[b][n, :n]).sum()) for b in range(B)]
        if any(nn):
            bad = np.argwhere(np.isnan(np.tril(Sm[int(np.argmax(nn))][:n + 1, :n])))
            print('   mode %d: NaNs in K^-1 per matrix %s; first at %s, last at %s (tile rows/cols /64: %s .. %s)' % (
                mode, nn, bad[0].tolist(), bad[-1].tolist(), (bad[0] // 64).tolist(), (bad[-1] // 64).tolist()))
    A0, l0, i0, S0, l20, i20, A20 = res[0]
    A1, l1, i1, S1, l21, i21, A21 = res[1]
    tl = np.tril_indices(n)
    eL = max(np.abs(A0[b][:n, :n][tl] - A1[b][:n, :n][tl]).max() for b in range(B))
    eR = max(np.abs(A0[b][n, :n + 1] - A1[b][n, :n + 1]).max() for b in range(B))
    eS = max(np.abs(np.tril(S0[b][:n, :n]) - np.tril(S1[b][:n, :n])).max() / np.abs(np.tril(S0[b][:n, :n])).max() for b in range(B))
    eA = max(np.abs(S0[b][n, :n] - S1[b][n, :n]).max() / (np.abs(S0[b][n, :n]).max() + 1e-300) for b in range(B))
    eld = np.abs(l0 - l1).max()
    good = eL < 1e-9 and eR < 1e-8 and eS < 1e-7 and eA < 1e-7 and eld < 1e-8 and not i0.any() and not i1.any() and not i21.any()
    ok &= bool(good)
    print('n=%4d B=%2d  |dL| %.1e  |d rhs row| %.1e  rel|dKinv| %.1e  rel|d alpha| %.1e  |d logdet| %.1e  info %s %s %s  %s' % (
        n, B, eL, eR, eS, eA, eld, i0.tolist()[:3], i1.tolist()[:3], i21.tolist()[:3], 'ok' if good else 'MISMATCH'))

# determinism under load: the same call 60 times, every output bitwise equal to the first
eng.set_potrf_mode(1)
for n, B, inv in ((2000, 6, True), (2000, 12, False), (1000, 16, True)):
    Np = eng.padded_dim(n)
    X, G, y = build(n, B, 5)
    work = eng.potrf_workspace(n, B)
    A = eng.empty(B, Np, Np)
    T, S = (eng.empty(B, Np, Np), eng.empty(B, Np, Np)) if inv else (None, None)
    ref = None
    ndiff = 0
    for rep in range(60):
        eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-4, out=A, full=False, Y=y, batch=B)
        if inv:
            ld, info = eng.potrf_inv(n, A, T, S, batch=B, work=work)
            out = torch.cat([torch.tril(A[:, :n + 1, :n + 1]).reshape(-1), torch.tril(S[:, :n + 1, :n + 1]).reshape(-1), ld])
        else:
            ld, info = eng.potrf(n, A, batch=B, work=work)
            out = torch.cat([torch.tril(A[:, :n + 1, :n + 1]).reshape(-1), ld])
        if ref is None:
            ref = out.clone()
        else:
            ndiff += int((out.view(torch.int64) != ref.view(torch.int64)).sum().item())
        if int(info.abs().sum().item()):
            print('info', info.tolist())
            ok = False
    print('determinism n=%d B=%d inv=%s: %d differing words over 59 repeats' % (n, B, inv, ndiff))
    ok &= ndiff == 0

# timings
ev0, ev1 = eng.event(), eng.event()
n = 2000
Np = eng.padded_dim(n)
print('n=%d   B | mode 0: potrf ms, potrf_inv ms | mode 1: potrf ms, potrf_inv ms (TF/s)' % n)
for B in (1, 2, 4, 6, 8, 12):
    X, G, y = build(n, B, 1)
    A = eng.empty(B, Np, Np)
    T, S = eng.empty(B, Np, Np), eng.empty(B, Np, Np)
    work = eng.potrf_workspace(n, B)
    row = []
    for mode in (0, 1):
        eng.set_potrf_mode(mode)
        tf, tv = [], []
        for rep in range(6):
            eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
            eng.record(ev0); eng.potrf(n, A, batch=B, work=work); eng.record(ev1)
            tf.append(eng.elapsed_ms(ev0, ev1))
            eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
            eng.record(ev0); eng.potrf_inv(n, A, T, S, batch=B, work=work); eng.record(ev1)
            tv.append(eng.elapsed_ms(ev0, ev1))
        row += [min(tf), min(tv)]
    print('        %2d | %7.3f %7.3f | %7.3f (%5.1f) %7.3f (%5.1f)' % (B, row[0], row[1], row[2], B * n ** 3 / 3 / row[2] / 1e9, row[3], B * n ** 3 / row[3] / 1e9))
print('OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
