"""cProfile of SI iterations at cfg4's shape (Vecchia DGP, n = 50 000, d = 8, m = 25): where the HOST time goes."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgp_amd import dgp
n, d, m = int(os.environ.get('N', '50000')), 8, 25
rng = np.random.default_rng(7)
X = rng.uniform(size=(n, d))
f = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2 + 0.3 * X[:, 3:].sum(1)
Y = ((f - f.mean()) / f.std())[:, None]
np.random.seed(0)
model = dgp(X, Y, vecchia=True, m=m, seed=1)
model.train(N=9, ess_burn=10, disable=True)   # (past the neighbour refreshes at 2, 4, 8)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
model.train(N=6, ess_burn=10, disable=True)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(45)
