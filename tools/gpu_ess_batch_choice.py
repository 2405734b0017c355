"""Number of ESS proposals an update really needs on the bench workload (sequential sampler, batch 1), and from it the
expected device time per update for speculative batch sizes (first, next) under the measured factorisation times."""
import sys, os, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

model, X, Y = bench.build_model(2000, 5, 100, 0)
imp = model.imp
imp.batch, imp.batch_next, imp.queued = 1, 1, False
need = []
for it in range(10):
    p0, u0 = imp.stats['proposals'], imp.stats['updates']
    # per update counts: run one sweep at a time
    for sweep in range(11):
        a, b = imp.stats['proposals'], imp.stats['updates']
        imp.sample(burnin=0)
        if it >= 1:
            need.append(imp.stats['proposals'] - a)
    model._m_step()
need = np.array(need)
print('updates', len(need), 'mean proposals needed %.2f' % need.mean())
h = collections.Counter(need.tolist())
cum = 0
for k in sorted(h):
    cum += h[k]
    print('%3d: %4d  cum %.3f' % (k, h[k], cum / len(need)))
# factorisation time (ms) of a batch of B matrices, n = 2000, measured here (one-launch kernel; HIP events, minimum of 5)
eng = model.engine
n = 2000
Np = eng.padded_dim(n)
rng = np.random.default_rng(0)
ev0, ev1 = eng.event(), eng.event()
tB = {}
import torch
for B in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16):
    Xb = eng.tensor(rng.uniform(size=(B, n, 5)))
    yb = eng.tensor(rng.normal(size=n))
    A = eng.empty(B, Np, Np)
    work = eng.potrf_workspace(n, B)
    ts = []
    for rep in range(6):
        eng.kmatrix('matern2.5', Xb, None, None, [1.0], 1e-6, out=A, full=False, Y=yb, batch=B)
        eng.record(ev0); eng.potrf(n, A, batch=B, work=work); eng.record(ev1)
        torch.cuda.synchronize()
        ts.append(eng.elapsed_ms(ev0, ev1))
    tB[B] = min(ts[1:])
print('potrf ms by batch:', ' '.join('%d:%.3f' % kv for kv in tB.items()))
def t(B):
    return tB[B] + 0.06 + 0.004 * B   # + K assembly and the small kernels
best = []
for b1 in tB:
    for b2 in tB:
        if b2 > b1:
            continue
        tot = 0.0
        for k in need:
            tot += t(b1)
            left = k - b1
            while left > 0:
                tot += t(b2)
                left -= b2
        best.append((tot / len(need), b1, b2))
best.sort()
for v, b1, b2 in best[:12]:
    print('first %2d next %2d: %.3f ms per update' % (b1, b2, v))
print('current 12/4: %.3f' % [v for v, b1, b2 in best if (b1, b2) == (12, 4)][0])
