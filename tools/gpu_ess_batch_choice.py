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
# factorisation time (ms) of a batch of B matrices, n = 2000 (profiles/r02_potrf_modes.txt, one-launch kernel, 6-panel visits)
tB = {1: 0.517, 2: 0.54, 3: 0.58, 4: 0.63, 5: 0.67, 6: 0.705, 7: 0.78, 8: 0.85, 9: 0.90, 10: 0.95, 11: 1.0, 12: 1.06, 14: 1.2, 16: 1.35}
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
