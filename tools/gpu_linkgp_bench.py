"""The linked-GP predictor alone (dgpamd_linkgp_predict) at a given shape, HIP-event timed, with the oracle's answer for
the first few test points.  usage: gpu_linkgp_bench.py kind n Dw Dz M   (cfg3's second layer: sexp 5000 10 10 2048)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine
kind = sys.argv[1] if len(sys.argv) > 1 else 'sexp'
n, Dw, Dz, M = (int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (5000, 10, 10, 2048)))
eng = Engine(0)
rng = np.random.default_rng(5)
W, Wg = rng.normal(size=(n, Dw)), (rng.uniform(size=(n, Dz)) if Dz else None)
if os.environ.get('ORDER'):   # training points grouped by cells (dgp_amd.ops.cell_order): what the emulator hands the Matern pair kernel
    from dgp_amd.ops import cell_order
    p = cell_order(W)
    W, Wg = W[p], (Wg[p] if Dz else None)
length, scale, nugget = np.array([2.5]), 1.3, 1e-4
# a symmetric "R^-1" and "R^-1 y" of plausible size (the kernel's time does not depend on their values)
G = rng.normal(size=(n, 8)) / np.sqrt(n)
Rinv = G @ G.T + np.eye(n)
ry = rng.normal(size=n)
m, v = rng.normal(size=(M, Dw)), rng.uniform(0.01, 0.4, size=(M, Dw))
z = rng.uniform(size=(M, Dz)) if Dz else None
dm, dv, dz = eng.tensor(m), eng.tensor(v), (eng.tensor(z) if Dz else None)
dW, dWg, dR, dry = eng.tensor(W), (eng.tensor(Wg) if Dz else None), eng.tensor(Rinv), eng.tensor(ry)


def run():
    return eng.linkgp_predict(kind, dm, dv, dz, dW, dWg, length, dR, n, dry, scale, nugget)


lm, lv = run()
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t = min(ts)
pairs = M * n * (n + 1) / 2
print('%s n=%d Dw=%d Dz=%d M=%d: %.1f ms -> %.0f pts/s, %.1f G pair-evaluations/s' % (kind, n, Dw, Dz, M, 1e3 * t, M / t, pairs / t / 1e9))
if os.environ.get('CHECK', '1') != '0':
    from oracle import dgp_oracle as O
    K = 3
    mo, vo = O.link_gp_predict(m[:K], v[:K], None if z is None else z[:K], W, Wg, Rinv, ry, scale, length, nugget, kind)
    a, b = lm[:K].cpu().numpy(), lv[:K].cpu().numpy()
    print('   vs oracle (first %d points): mean rel %.1e, var rel %.1e' % (K, np.max(np.abs(a - mo) / np.abs(mo)), np.max(np.abs(b - vo) / np.abs(vo))))
