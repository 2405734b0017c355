"""Sweep of the one-launch kernel's lazy depths (DGPAMD_MEGA_LAZY: panels per visit of an A/T tile, DGPAMD_MEGA_SLAZY: of a
K^-1 tile): correctness against the per-step kernel + timings; one process per setting (the task table is cached)."""
import os, subprocess, sys
if len(sys.argv) == 1:
    for lz, sz, cap in [tuple(int(v) for v in a.split(",")) for a in os.environ.get("SWEEP", "6,6,0 6,6,12").split()]:
        env = dict(os.environ, DGPAMD_MEGA_LAZY=str(lz), DGPAMD_MEGA_SLAZY=str(sz), DGPAMD_MEGA_CAP=str(cap))   # (CAP: experimental tables, not in the tree any more)
        r = subprocess.run([sys.executable, __file__, 'run'], env=env, capture_output=True, text=True, timeout=600)
        print('LAZY %d SLAZY %d CAP %d' % (lz, sz, cap)); print(r.stdout[-1500:], r.stderr[-300:] if r.returncode else '')
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgp_amd.ops import Engine
eng = Engine(0)
n = int(os.environ.get('N', 2000)); Np = eng.padded_dim(n)   # N=5000: cfg3's size
ev0, ev1 = eng.event(), eng.event()
for B in [int(v) for v in os.environ.get("BLIST", "1,3,6,12").split(",")]:
    r = np.random.default_rng(B)
    X = eng.tensor(r.uniform(size=(B, n, 5))); G = eng.tensor(r.uniform(size=(n, 5))); y = eng.tensor(r.normal(size=n))
    work = eng.potrf_workspace(n, B)
    out = {}
    tm = {}
    for mode in (0, 1):
        eng.set_potrf_mode(mode)
        A = eng.empty(B, Np, Np); T = eng.empty(B, Np, Np); S = eng.empty(B, Np, Np)
        tf, tv = [], []
        for rep in range(5):
            eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
            eng.record(ev0); ld, info = eng.potrf(n, A, batch=B, work=work); eng.record(ev1)
            tf.append(eng.elapsed_ms(ev0, ev1))
            L = torch.tril(A[:, :n + 1, :n]).clone()
            eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
            eng.record(ev0); ld2, info2 = eng.potrf_inv(n, A, T, S, batch=B, work=work); eng.record(ev1)
            tv.append(eng.elapsed_ms(ev0, ev1))
        out[mode] = (L, torch.tril(S[:, :n + 1, :n]).clone(), ld.clone(), ld2.clone(), int(info.abs().sum()) + int(info2.abs().sum()))
        tm[mode] = (min(tf), min(tv))
    dL = float((out[0][0] - out[1][0]).abs().max()); dS = float((out[0][1] - out[1][1]).abs().max() / out[0][1].abs().max())
    print('B=%2d  mode0 %.3f %.3f | mode1 %.3f %.3f ms   |dL| %.1e rel|dS| %.1e info %d %d' % (B, *tm[0], *tm[1], dL, dS, out[0][4], out[1][4]))
