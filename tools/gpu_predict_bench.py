"""Prediction leg only: emulator.predict on the bench model (2 SI iterations of training), timing + kernel check."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dgp_amd import emulator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
model, X, Y = bench.build_model(n, 5, 100, 0)
for _ in range(2):
    model.imp.sample(burnin=10)
    model._m_step()
model.N = max(model.N, 1)
emu = emulator(model.estimate(burnin=0), N=10)
Z = np.random.default_rng(5).uniform(size=(M, 5))
mu, var = emu.predict(Z)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); mu2, var2 = emu.predict(Z); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print('predict %d pts x 10 imputations: %.1f ms -> %.0f pts/s   mean[0:3]=%s var[0:3]=%s' % (M, 1e3 * min(ts), M / min(ts), mu[:3].ravel(), var[:3].ravel()))
