"""cProfile of emulator.predict at cfg4's shape (Vecchia DGP n = 50 000, 100 000 test points, 2 imputations): the HOST side."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dgp_amd import dgp, emulator
n, d, m = 50000, 8, 25
rng = np.random.default_rng(7)
X = rng.uniform(size=(n, d))
f = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2 + 0.3 * X[:, 3:].sum(1)
Y = ((f - f.mean()) / f.std())[:, None]
np.random.seed(0)
model = dgp(X, Y, vecchia=True, m=m, seed=1)
model.train(N=3, ess_burn=10, disable=True)
emu = emulator(model.estimate(), N=2, seed=3)
xt = rng.uniform(size=(int(os.environ.get('M', 100000)), d))
emu.predict(xt[:2000], m=50)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
mu, var = emu.predict(xt, m=50)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
