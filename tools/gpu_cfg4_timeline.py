"""Host timeline of ONE steady-state SI iteration of cfg4 training (Vecchia, n = 50 000), no extra synchronisation: entry / exit
times of the calls the iteration is made of (Engine.fetch = the host waits for the device)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd import dgp
from dgp_amd import imputation as I, kernel_class as K, mstep as MS, ops as O

n, d, m = int(os.environ.get('N', '50000')), 8, 25
rng = np.random.default_rng(7)
X = rng.uniform(size=(n, d))
f = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2 + 0.3 * X[:, 3:].sum(1)
Y = ((f - f.mean()) / f.std())[:, None]
np.random.seed(0)
model = dgp(X, Y, vecchia=True, m=m, seed=1)
warm = int(os.environ.get('WARM', '18'))
if warm:
    model.train(N=warm, ess_burn=10, disable=True)
log, depth = [], [0]


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def inner(*a, **k):
        t = time.perf_counter(); depth[0] += 1
        try:
            return fn(*a, **k)
        finally:
            depth[0] -= 1
            log.append((t, time.perf_counter(), depth[0], label))
    setattr(obj, name, inner)


imp, eng = model.imp, model.engine
for nm in ('one_sample_block', '_upper_loglik', 'update_ord_nn', '_attach', '_detach', 'finish_detach', '_prior_draws_ahead', '_vecchia_draws', 'stage_for_mstep', 'sample', '_sample_queued', '_queue_plan'):
    wrap(imp, nm)
for nm in ('vecchia_llik_batch', 'vecchia_llik', 'nn_ordered', 'collect', 'fetch', 'tensor', 'vecchia_lmatrix', 'vecchia_spsolve_levels', 'vecchia_levels'):
    wrap(eng, nm, 'eng.' + nm)
wrap(imp.draws, 'prefetch', 'draws.prefetch'); wrap(imp.draws, 'normals_device', 'draws.normals_device')
wrap(MS, 'maximise_lockstep_vecch'); wrap(MS, 'minimize_lockstep')
wrap(model, '_m_step'); wrap(model, '_fit_nodes'); wrap(model, '_si_iteration')
wrap(torch.Tensor, 'cpu', 'Tensor.cpu'); wrap(torch, 'stack', 'torch.stack'); wrap(torch, 'cat', 'torch.cat'); wrap(np.random, 'permutation', 'np.random.permutation'); wrap(np, 'argsort', 'np.argsort')
for nd in [x for layer in model.all_layer for x in layer if x.type == 'gp']:
    wrap(nd, 'r2'); wrap(nd, '_vecch_stage'); wrap(nd, '_opt_setup'); wrap(nd, 'ord_nn'); wrap(nd, 'nn_dev'); wrap(nd, 'ord_dev'); wrap(nd, '_X')
model.train(N=3, ess_burn=10, disable=True)
its = [e for e in log if e[3] == '_si_iteration']
pick = int(os.environ.get('PICK', '-1'))
t0, t1 = its[pick][0], its[pick][1]
print('iteration %.2f ms' % (1e3 * (t1 - t0)))
last = None
for a, b, dep, lab in sorted(log):
    if a < t0 or a > t1 or (b - a) < float(os.environ.get('MIN_US', '40')) * 1e-6:
        continue
    print('%9.3f .. %9.3f ms  (%7.3f)  %s%s' % (1e3 * (a - t0), 1e3 * (b - t0), 1e3 * (b - a), '  ' * dep, lab))
