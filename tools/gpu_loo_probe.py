import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from dgp_amd import dgp, kernel, combine, emulator
name = 'matern2.5'
for rep in range(int(os.environ.get('REPS', '4'))):
    rng = np.random.default_rng(5)
    if os.environ.get('SEEDNP'):
        np.random.seed(int(os.environ['SEEDNP']) + rep)
    n, d = 90, 2
    X = rng.uniform(size=(n, d))
    Y = np.sin(6.0 * X[:, [0]]) * np.cos(4.0 * X[:, [1]]) + 0.5 * X[:, [1]]
    layers = combine([kernel(length=np.array([1.0]), name=name) for _ in range(d)],
                     [kernel(length=np.array([1.0]), name=name, connect=np.arange(d)) for _ in range(d)],
                     [kernel(length=np.array([1.0]), name=name, scale_est=True)])
    model = dgp(X, Y, layers, seed=2)
    model.train(N=6, ess_burn=3, disable=True)
    hyp = np.concatenate([np.concatenate((nd.scale, nd.length)) for layer in model.all_layer for nd in layer])
    emu = emulator(model.estimate(), N=3)
    mu, var = emu.loo(X)
    gps = [nd for layer in emu.all_layer for nd in layer if nd.type == 'gp']
    for nd in gps:
        nd.loo_state, nd.vecch = True, True
    try:
        mu_ref, var_ref = emu._predict_vecchia(X, False, n, True)
    finally:
        for nd in gps:
            nd.loo_state, nd.vecch = False, False
    dv = np.abs(var - var_ref); tol = 1e-6 + 1e-5 * np.abs(var_ref)
    print('rep %d: hyper checksum %.15g | max |dvar| %.3e (worst excess over tolerance %.3e) | max |dmu| %.3e' % (rep, hyp.sum(), dv.max(), (dv - tol).max(), np.abs(mu - mu_ref).max()), flush=True)
