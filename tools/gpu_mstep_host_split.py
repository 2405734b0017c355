"""Where the wall time of the lock-step M-step goes: device round trips (dgpamd_llik_batch incl. its one sync) against
the host work between them (scipy's L-BFGS-B core, node updates, result unpacking).  cfg2 shape.
Run on the GPU box: python tools/gpu_mstep_host_split.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dgp_amd import mstep  # noqa: E402

model, X, Y = bench.build_model(2000, 5, 100, 0)
for _ in range(3):
    model.imp.sample(burnin=10)
    model._m_step()

t_run = [0.0]
n_run = [0]
orig_plan = model.engine.llik_plan


def plan_wrap(*a, **k):
    p = orig_plan(*a, **k)
    run = p.run

    def timed(pos):
        t0 = time.perf_counter()
        r = run(pos)
        t_run[0] += time.perf_counter() - t0
        n_run[0] += 1
        return r
    p.run = timed
    return p


model.engine.llik_plan = plan_wrap
tot_i = tot_m = 0.0
K = 10
for _ in range(K):
    t0 = time.perf_counter()
    model.imp.sample(burnin=10)
    model.engine.sync()
    t1 = time.perf_counter()
    model._m_step()
    model.engine.sync()
    t2 = time.perf_counter()
    tot_i += t1 - t0
    tot_m += t2 - t1
print('per SI iteration: I-step %.2f ms, M-step %.2f ms' % (1e3 * tot_i / K, 1e3 * tot_m / K))
print('M-step: %.1f device round trips of %.3f ms = %.2f ms; host between them %.2f ms (%.0f us per round)'
      % (n_run[0] / K, 1e3 * t_run[0] / n_run[0], 1e3 * t_run[0] / K, 1e3 * (tot_m - t_run[0]) / K,
         1e6 * (tot_m - t_run[0]) / n_run[0]))
