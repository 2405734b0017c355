import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import torch
from dgp_amd.ops import Engine, cell_order
eng = Engine(0)
rng = np.random.default_rng(5)
n, Dw, M = 2000, 5, 2048
W = rng.normal(size=(n, Dw)); W = W[cell_order(W)]
G = rng.normal(size=(n, 8)) / np.sqrt(n); Rinv = G @ G.T + np.eye(n); ry = rng.normal(size=n)
m, v = rng.normal(size=(M, Dw)), rng.uniform(0.01, 0.4, size=(M, Dw))
dm, dv, dW, dR, dry = eng.tensor(m), eng.tensor(v), eng.tensor(W), eng.tensor(Rinv), eng.tensor(ry)
length = np.array([2.5])
run = lambda: [t.cpu().numpy() for t in eng.linkgp_predict('matern2.5', dm, dv, None, dW, None, length, dR, n, dry, 1.3, 1e-4)]
eng.set_linkgp_direct(True); m0, v0 = run(); eng.set_linkgp_direct(False)
for rep in range(3):
    m1, v1 = run()
    bad = np.where(np.abs(v1 - v0) > 1e-6 * np.abs(v0))[0]
    print('rep', rep, 'bad points:', len(bad), bad[:40], 'mod 32:', sorted(set((bad % 32).tolist()))[:10])
