"""Scale probes for BASELINE configs 3 and 4 (run on the GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synthetic
from dgp_amd import dgp, kernel, combine, emulator
from dgp_amd.ops import default_engine

which = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
eng = default_engine(0)


def sync():
    torch.cuda.synchronize()


if which == 'cfg3':
    n, d, q = int(os.environ.get('N', '5000')), 10, 3
    rng = np.random.default_rng(2026)
    X = rng.uniform(size=(n, d))
    Y = np.stack([np.sin(1.0 / ((0.7 * X[:, 0] + 0.3) * (0.7 * X[:, 1] + 0.3))) + (0.2 + 0.1 * j) * (X[:, 2 + j:] ** 2).sum(1) for j in range(q)], 1)
    Y = (Y - Y.mean(0)) / Y.std(0)
    t = time.perf_counter()
    model = dgp(X, Y, seed=1)          # default structure: d SExp nodes (shared lengthscale) -> q SExp nodes with global connection
    sync(); print('cfg3 n=%d: construct (11 sweeps) %.2f s' % (n, time.perf_counter() - t))
    if os.environ.get('BATCH'):   # (speculative batch sizes of the I-step: first, later, batches queued per update)
        model.imp.batch, model.imp.batch_next = int(os.environ['BATCH']), int(os.environ.get('BATCH_NEXT', os.environ['BATCH']))
        model.imp.queue_max_batches = int(os.environ.get('QMAX', '2'))
        model.imp._batch_default = False
    its = int(os.environ.get('ITERS', '2'))
    t = time.perf_counter(); model.train(N=its, ess_burn=10, disable=True); sync()
    print('cfg3: %d SI iterations %.2f s -> %.2f it/s; stats %s' % (its, time.perf_counter() - t, its / (time.perf_counter() - t), model.imp.stats))
    if os.environ.get('TRAIN_ONLY'):
        sys.exit(0)
    S, M = int(os.environ.get('S', '6')), int(os.environ.get('M', '4096'))
    t = time.perf_counter(); emu = emulator(model.estimate(burnin=0), N=S, seed=3); sync()
    print('cfg3: emulator(N=%d) %.2f s' % (S, time.perf_counter() - t))
    xt = rng.uniform(size=(M, d))
    t = time.perf_counter(); emu.predict(xt[:64]); sync(); print('cfg3: stats + first predict %.2f s' % (time.perf_counter() - t))
    t = time.perf_counter(); mu, var = emu.predict(xt); sync(); dt = time.perf_counter() - t
    print('cfg3: predict %d pts x %d imputations %.2f s -> %.0f pts/s (x%d imputations = %.0f pt-imputations/s); finite=%s; mem %.1f GB'
          % (M, S, dt, M / dt, S, M * S / dt, np.all(np.isfinite(mu)) and np.all(np.isfinite(var)), torch.cuda.max_memory_allocated() / 2**30))
elif which == 'cfg5':
    # BASELINE config 5: GP -> DGP -> GP feed-forward chain, n = 1000 each, Matern-2.5, predict 1e4 points through lgp
    from dgp_amd import gp, lgp, container
    n, M = int(os.environ.get('N', '1000')), int(os.environ.get('M', '10000'))
    rng = np.random.default_rng(5)
    X1 = rng.uniform(size=(n, 3))
    Y1 = np.sin(3 * X1[:, :1]) + X1[:, 1:2] ** 2 - X1[:, 2:]
    Y1 = (Y1 - Y1.mean()) / Y1.std()
    t = time.perf_counter()
    g1 = gp(X1, Y1, kernel(length=np.array([0.8, 1.2, 1.0]), name='matern2.5', scale_est=True, nugget=1e-4)); g1.train()
    Y2 = np.tanh(2 * Y1) + 0.3 * Y1 ** 2
    Y2 = (Y2 - Y2.mean()) / Y2.std()
    d2 = dgp(Y1, Y2, combine([kernel(length=np.array([1.0]), name='matern2.5')],
                             [kernel(length=np.array([1.0]), name='matern2.5', scale_est=True, connect=np.arange(1))]), seed=2)
    d2.train(N=20, ess_burn=10, disable=True)
    Y3 = np.cos(2 * Y2)
    Y3 = (Y3 - Y3.mean()) / Y3.std()
    g3 = gp(Y2, Y3, kernel(length=np.array([1.0]), name='matern2.5', scale_est=True, nugget=1e-4)); g3.train()
    sync(); print('cfg5 n=%d: three emulators trained (GP, 20 SI iterations of the DGP, GP) in %.1f s' % (n, time.perf_counter() - t), flush=True)
    sysm = lgp(combine([container(g1.export(), np.array([0, 1, 2]))], [container(d2.estimate(), np.array([0]))],
                       [container(g3.export(), np.array([0]))]), N=10)
    xt = rng.uniform(size=(M, 3))
    mu, var = sysm.predict([xt[:64], [None], [None]]); sync()
    t = time.perf_counter(); mu, var = sysm.predict([xt, [None], [None]]); sync(); dt = time.perf_counter() - t
    y1 = np.sin(3 * xt[:, :1]) + xt[:, 1:2] ** 2 - xt[:, 2:]
    print('cfg5: lgp.predict %d points x 10 imputations %.2f s -> %.0f pts/s; finite %s' % (M, dt, M / dt, bool(np.all(np.isfinite(mu[0])) and np.all(np.isfinite(var[0])))))
elif which == 'cfg4train':
    # BASELINE config 4 through the public API: Vecchia DGP (default 2-layer structure), a few SI iterations + prediction
    n, d, m = int(os.environ.get('N', '50000')), 8, 25
    rng = np.random.default_rng(7)
    X = rng.uniform(size=(n, d))
    f = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2 + 0.3 * X[:, 3:].sum(1)
    Y = ((f - f.mean()) / f.std())[:, None]
    np.random.seed(0)   # (the Vecchia ordering is a numpy.random.permutation: the same work in every run)
    t = time.perf_counter(); model = dgp(X, Y, vecchia=True, m=m, seed=1); sync()
    print('cfg4train n=%d: construct (warm start + NN + 11 sweeps) %.1f s' % (n, time.perf_counter() - t), flush=True)
    its = int(os.environ.get('ITERS', '3'))
    if os.environ.get('BATCH'):
        model.imp.batch = int(os.environ['BATCH'])
        model.imp.batch_next = int(os.environ.get('BATCH_NEXT', os.environ['BATCH']))
    if os.environ.get('QMAX'):
        model.imp.queue_max_batches = int(os.environ['QMAX'])
    if os.environ.get('NOQUEUE'):
        model.imp.queued = False   # (the host-driven accept / shrink loop: one synchronisation per speculative batch)
    spent, inner = [0.0], model._m_step

    def timed_m_step(*a, **k):
        sync(); t0 = time.perf_counter()
        r = inner(*a, **k)
        sync(); spent[0] += time.perf_counter() - t0
        return r
    model._m_step = timed_m_step
    if os.environ.get('ITER_TIMES'):   # wall time of every iteration (the neighbour refreshes at 2, 4, 8, .. stand out)
        it_t, inner_it = [], model._si_iteration

        def timed_it(*a, **k):
            t0 = time.perf_counter()
            try:
                return inner_it(*a, **k)
            finally:
                it_t.append(time.perf_counter() - t0)
        model._si_iteration = timed_it
    t = time.perf_counter(); model.train(N=its, ess_burn=10, disable=True); sync(); dt = time.perf_counter() - t
    if os.environ.get('ITER_TIMES'):
        print('cfg4train: ms per iteration ' + ' '.join('%.0f' % (1e3 * v) for v in it_t), flush=True)
    print('cfg4train: %d SI iterations %.1f s -> %.3f it/s (M-steps %.0f ms each, the rest %.0f ms); stats %s'
          % (its, dt, its / dt, 1e3 * spent[0] / its, 1e3 * (dt - spent[0]) / its, model.imp.stats), flush=True)
    if os.environ.get('TRAIN_ONLY'):
        sys.exit(0)
    t = time.perf_counter(); emu = emulator(model.estimate(burnin=0), N=2, seed=3); sync()
    print('cfg4train: emulator(N=2) %.1f s' % (time.perf_counter() - t), flush=True)
    for mp in [int(v) for v in os.environ.get('MPRED', '2000').split(',')]:
        xt = rng.uniform(size=(mp, d))
        ft = np.sin(3 * xt[:, 0]) * np.cos(2 * xt[:, 1]) + xt[:, 2] ** 2 + 0.3 * xt[:, 3:].sum(1)
        t = time.perf_counter(); mu, var = emu.predict(xt, m=50); sync(); dt = time.perf_counter() - t
        print('cfg4train: predict %d pts x 2 imputations %.2f s (%.0f pts/s), rmse %.3f, mem %.1f GB' % (
            mp, dt, mp / dt, np.sqrt(np.mean((mu[:, 0] - (ft - f.mean()) / f.std()) ** 2)), torch.cuda.max_memory_allocated() / 2**30))
else:
    n, d, m = int(os.environ.get('N', '50000')), 8, 25
    rng = np.random.default_rng(7)
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 + 0.1 * rng.normal(size=n)
    length = np.full(d, 0.8)
    Xs = eng.tensor(X / length)
    t = time.perf_counter(); NN = eng.nn_ordered(Xs, m); sync(); t_nn = time.perf_counter() - t
    print('cfg4 n=%d: ordered %d-NN %.3f s (%.0f rows/s)' % (n, m, t_nn, n / t_nn))
    dX, dy, ones = eng.tensor(X), eng.tensor(y), eng.tensor(np.ones(n))
    for name in ('sexp', 'matern2.5'):
        for rep in range(2):
            t = time.perf_counter(); out = eng.vecchia_llik(name, dX, dy, NN, length, 1e-4, ones); sync(); t1 = time.perf_counter() - t
            t = time.perf_counter(); o, P = eng.vecchia_nllik(name, dX, dy, NN, length, 1e-4, ones, True); sync(); t2 = time.perf_counter() - t
            t = time.perf_counter(); Lm = eng.vecchia_lmatrix(name, dX, NN, length, 1e-4); sync(); t3 = time.perf_counter() - t
            t = time.perf_counter(); xs = eng.vecchia_spsolve(Lm, NN, 1.0, eng.tensor(rng.normal(size=n))); sync(); t4 = time.perf_counter() - t
        print('cfg4 %s: llik %.1f ms (%.2f Mrows/s) | nllik(P=%d) %.1f ms (%.2f Mrows/s) | L_matrix %.1f ms | sparse solve %.1f ms | finite %s'
              % (name, 1e3 * t1, n / t1 / 1e6, P, 1e3 * t2, n / t2 / 1e6, 1e3 * t3, 1e3 * t4, bool(torch.isfinite(out).all() and torch.isfinite(xs).all())))
    xq = rng.uniform(size=(10000, d))
    t = time.perf_counter(); PN = eng.nn_query(eng.tensor(xq / length), Xs, 50); sync(); t5 = time.perf_counter() - t
    t = time.perf_counter(); gm, gv = eng.vecchia_gp('matern2.5', eng.tensor(xq), dX, PN, dy, 1.0, length, 1e-4, ones); sync(); t6 = time.perf_counter() - t
    print('cfg4: pred NN (50 of %d) for 1e4 queries %.1f ms | gp_vecch 1e4 pts %.1f ms (%.0f pts/s)' % (n, 1e3 * t5, 1e3 * t6, 1e4 / t6))
