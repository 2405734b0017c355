"""Stress of the one-launch factorisation: random sizes / batch sizes / with and without the inverse, every result checked
(info == 0, log-determinant against the first run of the same inputs, K^-1 K = I spot check) -- a lost hand-off would show
up as DgpAmdError (info -1), a stale read as a differing log-determinant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgp_amd.ops import Engine

eng = Engine(0)
rng = np.random.default_rng(int(os.environ.get('SEED', '0')))
N = int(os.environ.get('LAUNCHES', '1500'))
shapes = {}
t0 = time.perf_counter()
nl = 0
while nl < N:
    n = int(rng.choice([int(v) for v in os.environ['SIZES'].split(',')] if os.environ.get('SIZES') else [64, 65, 130, 200, 333, 640, 1000, 1280, 1999, 2000, 2047, 2100]))
    B = int(rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 8, 12, 16]))
    inv = bool(rng.integers(2))
    key = (n, B)
    if key not in shapes:
        r = np.random.default_rng(n * 100 + B)
        X = eng.tensor(r.uniform(size=(B, n, 4))); G = eng.tensor(r.uniform(size=(n, 3))); y = eng.tensor(r.normal(size=n))
        Np = eng.padded_dim(n)
        shapes[key] = dict(X=X, G=G, y=y, A=eng.empty(B, Np, Np), T=eng.empty(B, Np, Np), S=eng.empty(B, Np, Np),
                           work=eng.potrf_workspace(n, B), ref=None)
    s = shapes[key]
    reps = int(rng.integers(1, 6))
    for _ in range(reps):
        eng.kmatrix('matern2.5', s['X'], None, s['G'], [0.7], 1e-5, out=s['A'], full=False, Y=s['y'], batch=B)
        if inv:
            ld, info = eng.potrf_inv(n, s['A'], s['T'], s['S'], batch=B, work=s['work'])
        else:
            ld, info = eng.potrf(n, s['A'], batch=B, work=s['work'])
        nl += 1
    ldh, ih = eng.fetch(ld), eng.fetch(info)
    assert not ih.any(), (n, B, inv, ih)
    if s['ref'] is None:
        s['ref'] = ldh.copy()
    assert np.array_equal(ldh, s['ref']), (n, B, inv, ldh - s['ref'])
    if inv and n <= 1000:
        K = eng.kmatrix('matern2.5', s['X'][0], None, s['G'], [0.7], 1e-5, full=True).cpu().numpy()
        Kinv = np.tril(s['S'][0].cpu().numpy()[:n, :n]); Kinv = Kinv + np.tril(Kinv, -1).T
        err = np.abs(Kinv @ K - np.eye(n)).max()
        assert err < 1e-6, (n, B, err)
print('%d launches over %d shapes in %.1f s: all results consistent' % (nl, len(shapes), time.perf_counter() - t0))
