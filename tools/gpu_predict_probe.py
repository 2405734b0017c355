"""Prediction-leg probe at the bench shapes (n=2000, d=5, Matern): emulator build + predict timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from dgp_amd import emulator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
model, X, Y = build_model(n, 5, 100, 0)
model.train(N=2, ess_burn=5, disable=True)
t0 = time.perf_counter()
emu = emulator(model.estimate(burnin=0), N=10, seed=7)
t1 = time.perf_counter()
xt = np.random.default_rng(5).uniform(size=(M, 5))
emu.predict(xt[:32])
torch.cuda.synchronize()
t2 = time.perf_counter()
for rep in range(2):
    mu, var = emu.predict(xt)
torch.cuda.synchronize()
t3 = time.perf_counter()
print('emulator build %.2f s | stats+first predict %.2f s | predict %d pts x 10 imputations: %.3f s -> %.0f pts/s'
      % (t1 - t0, t2 - t1, M, (t3 - t2) / 2, M / ((t3 - t2) / 2)))
