"""The one-launch factorisation at 16-64 matrices per call against the per-block-step kernel (bit-identical factors, inverses, log-determinants): the critical-lane workers are capped per XCD so that bulk-first workers remain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgp_amd.ops import Engine
eng = Engine(0)
for n in (300, 1000):
    Np = eng.padded_dim(n)
    for B in (16, 24, 32, 48, 64):
        r = np.random.default_rng(B)
        X = eng.tensor(r.uniform(size=(B, n, 4))); G = eng.tensor(r.uniform(size=(n, 3))); y = eng.tensor(r.normal(size=n))
        work = eng.potrf_workspace(n, B)
        out = {}
        for mode in (0, 1):
            eng.set_potrf_mode(mode)
            A = eng.empty(B, Np, Np); T = eng.empty(B, Np, Np); S = eng.empty(B, Np, Np)
            for rep in range(3):
                eng.kmatrix('matern2.5', X, None, G, [0.7], 1e-5, out=A, full=False, Y=y, batch=B)
                ld, info = eng.potrf(n, A, batch=B, work=work)
                L = torch.tril(A[:, :n + 1, :n]).clone()
                eng.kmatrix('matern2.5', X, None, G, [0.7], 1e-5, out=A, full=False, Y=y, batch=B)
                ld2, info2 = eng.potrf_inv(n, A, T, S, batch=B, work=work)
            torch.cuda.synchronize()
            out[mode] = (L, torch.tril(S[:, :n + 1, :n]).clone(), ld.clone(), ld2.clone(), int(info.abs().sum()) + int(info2.abs().sum()))
        dL = float((out[0][0] - out[1][0]).abs().max()); dS = float((out[0][1] - out[1][1]).abs().max())
        print('n=%d B=%d: |dL| %.1e |dS| %.1e dlogdet %.1e info %d %d' % (n, B, dL, dS, float((out[0][2]-out[1][2]).abs().max()), out[0][4], out[1][4]), flush=True)
