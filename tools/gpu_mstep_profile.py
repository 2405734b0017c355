"""Where an M-step's wall time goes: rounds, device pipeline (incl. sync), host optimiser work."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from dgp_amd import mstep

model, X, Y = build_model(2000, 5, 100, 0)
for _ in range(2):
    model.imp.sample(burnin=10); model._m_step()
from dgp_amd import ops
T = dict(dev=0.0, n=0)
orig = ops._LlikPlan.run
def timed(self, idx):
    t = time.perf_counter(); r = orig(self, idx); T['dev'] += time.perf_counter() - t; T['n'] += 1; return r
ops._LlikPlan.run = timed
N = 6
tot = 0.0
rounds = evals = 0
for _ in range(N):
    model.imp.sample(burnin=10)
    torch.cuda.synchronize(); t = time.perf_counter(); model._m_step(); torch.cuda.synchronize(); tot += time.perf_counter() - t
    rounds += model.last_mstep[0]; evals += model.last_mstep[1]
print('M-step %.1f ms: %.1f rounds, %.1f evals; device pipeline calls %.1f x %.2f ms = %.1f ms; host rest %.1f ms'
      % (1e3 * tot / N, rounds / N, evals / N, T['n'] / N, 1e3 * T['dev'] / max(1, T['n']), 1e3 * T['dev'] / N, 1e3 * (tot - T['dev']) / N))
import cProfile, pstats
model.imp.sample(burnin=10)
pr = cProfile.Profile(); pr.enable(); model._m_step(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
