"""Exercise the public constructors the way the reference's demos do (default structures, options); prints one line
per scenario.  Run on the GPU box: python tools/gpu_api_tour.py"""
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgp_amd import dgp, gp, kernel, combine, emulator, lgp, container  # noqa: E402

rng = np.random.default_rng(0)
X = rng.uniform(size=(40, 2))
Y = np.sin(6 * X[:, [0]]) * X[:, [1]]
Xr = np.repeat(X[:15], 3, axis=0)
Yr = np.sin(6 * Xr[:, [0]]) + 0.1 * rng.normal(size=(45, 1))
Yc = rng.poisson(np.exp(1 + np.sin(4 * X[:, [0]])))
Ycat = (X[:, [0]] > 0.5).astype(int) + (X[:, [1]] > 0.5).astype(int)
Y2 = np.hstack((Y, np.cos(3 * X[:, [0]])))

from dgp_amd import Hetero, Poisson, NegBin, Categorical  # noqa: E402


def K(name='sexp', D=1, **kw):
    return kernel(length=np.ones(D), name=name, **kw)


def two(name='sexp', top=None, **kw):
    return combine([K(name, **kw) for _ in range(2)], [top if top is not None else K(name, scale_est=True, connect=np.arange(2), **kw)])


scen = {
    'default structure': lambda: dgp(X, Y),
    'default, block=False': lambda: dgp(X, Y, block=False),
    'depth 3 sexp': lambda: dgp(X, Y, combine([K() for _ in range(2)], [K(connect=np.arange(2)) for _ in range(2)],
                                              [K(scale_est=True, connect=np.arange(2))])),
    'matern, per-dim lengths, nugget_est': lambda: dgp(X, Y, combine(
        [K('matern2.5', 2, nugget_est=True, nugget=1e-3) for _ in range(2)],
        [K('matern2.5', 4, nugget_est=True, nugget=1e-3, scale_est=True, connect=np.arange(2))])),
    'prior ref': lambda: dgp(X, Y, two(prior_name='ref')),
    'prior inv_ga + bds': lambda: dgp(X, Y, two(prior_name='inv_ga', bds=np.array([0.05, 20.0]))),
    'prior None': lambda: dgp(X, Y, two(prior_name=None)),
    'no connect': lambda: dgp(X, Y, two(top=K(scale_est=True))),
    'input_dim subsets': lambda: dgp(X, Y, combine([K(input_dim=np.array([0])), K(input_dim=np.array([1]))],
                                                   [K(scale_est=True, input_dim=np.array([0, 1]))])),
    'two outputs': lambda: dgp(X, Y2, combine([K() for _ in range(2)], [K(scale_est=True, connect=np.arange(2)) for _ in range(2)])),
    'replicates + Hetero': lambda: dgp(Xr, Yr, combine([K() for _ in range(2)], [K(scale_est=True), K(scale_est=True)],
                                                        [Hetero()])),
    'Poisson': lambda: dgp(X, Yc.astype(float), combine([K() for _ in range(2)], [K(scale_est=True)], [Poisson()])),
    'NegBin': lambda: dgp(X, Yc.astype(float), combine([K() for _ in range(2)], [K(scale_est=True), K(scale_est=True)], [NegBin()])),
    'Categorical 3 classes': lambda: dgp(X, Ycat, combine([K() for _ in range(2)], [K(scale_est=True) for _ in range(3)],
                                                          [Categorical(num_classes=3)])),
    'vecchia m=10': lambda: dgp(X, Y, vecchia=True, m=10),
    'vecchia + ord_fun': lambda: dgp(X, Y, vecchia=True, m=10, ord_fun=lambda x: np.argsort(x[:, 0])),
    'vecchia Hetero': lambda: dgp(Xr, Yr, combine([K() for _ in range(2)], [K(scale_est=True), K(scale_est=True)], [Hetero()]),
                                  vecchia=True, m=10),
    'check_rep=False': lambda: dgp(Xr, Yr, check_rep=False),
}
for name, make in scen.items():
    try:
        m = make()
        m.train(N=3, ess_burn=2, disable=True)
        emu = emulator(m.estimate(), N=2)
        out = emu.predict(X[:5])
        ok = all(np.all(np.isfinite(np.asarray(o))) for o in out)
        print('%-34s layers %s  predict %s finite %s' % (name, [len(l) for l in m.all_layer], np.asarray(out[0]).shape, ok))
    except Exception:
        print('%-34s FAILED' % name)
        traceback.print_exc()

try:
    g1 = gp(X, Y, kernel(length=np.array([0.5, 0.5]), name='sexp', scale_est=True))
    g1.train()
    g2 = gp(X, Y, kernel(length=np.array([0.5]), name='matern2.5', scale_est=True), vecchia=True, m=8)
    g2.train()
    print('gp dense / vecchia predict', g1.predict(X[:3])[0].ravel(), g2.predict(X[:3])[0].ravel())
    print('gp export ->', type(g1.export()).__name__)
except Exception:
    print('gp FAILED')
    traceback.print_exc()
