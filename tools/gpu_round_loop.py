"""Loop one M-step round (dgpamd_llik_batch) at n = 2000 for a kernel trace.  usage: gpu_round_loop.py B reps"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine
eng = Engine(0)
n = 2000
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(0)
specs = [dict(kind='matern2.5', Xloc=eng.tensor(rng.uniform(size=(n, 5))), Xglob=None, nlen=1, nugget_est=False, W=None,
              y=eng.tensor(rng.normal(size=n))) for _ in range(B)]
plan = eng.llik_plan(n, specs)
for b in range(B):
    plan.set(b, [1.0], 1e-6)
idx = list(range(B))
for _ in range(5):
    plan.run(idx)
t = time.perf_counter()
for _ in range(reps):
    plan.run(idx)
print('B=%d %s: %.1f us per round' % (B, 'five launches' if os.environ.get('DGPAMD_LLIK_UNFUSED') else 'three launches', 1e6 * (time.perf_counter() - t) / reps))
