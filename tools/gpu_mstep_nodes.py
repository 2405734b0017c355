"""Per-node objective evaluations of the lock-step M-step at the bench shape: who is the slow node, how many matrices
does a round carry, how long does a round of B matrices take."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from dgp_amd import mstep, ops

model, X, Y = build_model(2000, 5, 0, 0)
for _ in range(3):
    model.imp.sample(burnin=10); model._m_step()

log = []
orig_run = ops._LlikPlan.run
def timed(self, idx):
    torch.cuda.synchronize(); t = time.perf_counter(); r = orig_run(self, idx); dt = time.perf_counter() - t
    log.append((len(idx), tuple(idx), dt)); return r
ops._LlikPlan.run = timed
hist = {}
per_node = []
for it in range(12):
    model.imp.sample(burnin=10)
    log.clear()
    model._m_step()
    cnt = {}
    for B, idx, dt in log:
        for i in idx:
            cnt[i] = cnt.get(i, 0) + 1
        hist.setdefault(B, []).append(dt)
    per_node.append([cnt.get(i, 0) for i in range(6)])
    print('iter %2d rounds %2d evals per node (pos 0-4 layer 1, 5 top)' % (it, len(log)), per_node[-1], 'B per round', [B for B, _, _ in log])
print('mean evals per node', np.mean(per_node, 0))
for B in sorted(hist):
    print('B=%d: %4d rounds, %.3f ms per round (call incl. sync)' % (B, len(hist[B]), 1e3 * np.mean(hist[B])))
