"""cProfile of the host side of cfg4 training (Vecchia, n = 50 000): which Python functions the wall time of an SI iteration
goes to (the engine's fetch = waiting for the device)."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd import dgp

n, d, m = int(os.environ.get('N', '50000')), 8, 25
rng = np.random.default_rng(7)
X = rng.uniform(size=(n, d))
f = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2 + 0.3 * X[:, 3:].sum(1)
Y = ((f - f.mean()) / f.std())[:, None]
np.random.seed(0)
model = dgp(X, Y, vecchia=True, m=m, seed=1)
model.train(N=17, ess_burn=10, disable=True)   # (past the neighbour refreshes at 2, 4, 8, 16)
its = int(os.environ.get('ITERS', '14'))
pr = cProfile.Profile()
torch.cuda.synchronize(); t0 = time.perf_counter()
pr.enable()
model.train(N=its, ess_burn=10, disable=True)
pr.disable()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('%d iterations, %.1f ms each under cProfile' % (its, 1e3 * dt / its))
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(int(os.environ.get('TOP', '40')))
if os.environ.get('CUM'):
    st.sort_stats('cumulative').print_stats(45)
