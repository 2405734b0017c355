"""Soak: a longer training run at the bench shape + emulator + predictions; checks finiteness and timing drift."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from dgp_amd import emulator

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
model, X, Y = build_model(2000, 5, 100, 0)
if os.environ.get('ESS_BATCH'):
    model.imp.batch = int(os.environ['ESS_BATCH'])
if os.environ.get('ESS_BATCH_NEXT'):
    model.imp.batch_next = int(os.environ['ESS_BATCH_NEXT'])
t0 = time.perf_counter()
ts = []
for blk in range(N // 10):
    t = time.perf_counter(); model.train(N=10, ess_burn=10, disable=True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 10)
print('train %d iterations: %.2f s; ms/iter per block of 10: %s' % (N, time.perf_counter() - t0, ' '.join('%.1f' % (1e3 * v) for v in ts)))
for l, layer in enumerate(model.all_layer):
    for k, nd in enumerate(layer):
        pp = nd.para_path
        assert np.all(np.isfinite(pp)) and pp.shape[0] == N + 1, (l, k, pp.shape)
print('layer-2 node: scale %.4g length %s nugget %.3g' % (model.all_layer[1][0].scale[0], model.all_layer[1][0].length, model.all_layer[1][0].nugget[0]))
print('layer-1 lengths:', [float(nd.length[0]) for nd in model.all_layer[0]])
emu = emulator(model.estimate(), N=10, seed=3)
xt = np.random.default_rng(9).uniform(size=(512, 5))
from bench import synthetic
mu, var = emu.predict(xt)
f = np.sin(1.0 / ((0.7 * xt[:, 0] + 0.3) * (0.7 * xt[:, 1] + 0.3)))
for k in range(2, 5):
    f = f + (0.3 + 0.2 * k) * xt[:, k] ** 2
Xtr, Ytr = synthetic(2000, 5)
ftr = np.sin(1.0 / ((0.7 * Xtr[:, 0] + 0.3) * (0.7 * Xtr[:, 1] + 0.3)))
for k in range(2, 5):
    ftr = ftr + (0.3 + 0.2 * k) * Xtr[:, k] ** 2
ytrue = (f - ftr.mean()) / ftr.std()
rmse = float(np.sqrt(np.mean((mu[:, 0] - ytrue) ** 2)))
cover = float(np.mean(np.abs(mu[:, 0] - ytrue) < 2 * np.sqrt(np.maximum(var[:, 0], 0))))
print('held-out RMSE %.4f (output standardised), 2-sigma coverage %.2f, var range [%.2e, %.2e], stats %s' % (rmse, cover, var.min(), var.max(), model.imp.stats))
