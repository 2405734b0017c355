"""Host-side phase times of one SI iteration at BASELINE configs[3] (Vecchia, n = 50 000): which calls of the I-step and the
M-step the wall time goes to (device synchronised around each)."""
import os, sys, time, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd import dgp
from dgp_amd import imputation as I, kernel_class as K, mstep as MS

n, d, m = int(os.environ.get('N', '50000')), 8, 25
rng = np.random.default_rng(7)
X = rng.uniform(size=(n, d))
f = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2 + 0.3 * X[:, 3:].sum(1)
Y = ((f - f.mean()) / f.std())[:, None]
np.random.seed(0)
model = dgp(X, Y, vecchia=True, m=m, seed=1)
model.train(N=3, ess_burn=10, disable=True)
acc = collections.defaultdict(float)
cnt = collections.Counter()


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def inner(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[label] += time.perf_counter() - t; cnt[label] += 1
        return r
    setattr(obj, name, inner)


imp = model.imp
for nm in ('_attach', '_detach', '_prior_draws_ahead', 'one_sample_block', '_vecchia_draws', '_upper_loglik', 'stage_for_mstep', 'sample', '_sample_queued', '_queue_plan'):
    wrap(imp, nm)
wrap(imp.draws, 'normals', 'draws.normals')
wrap(MS, 'maximise_lockstep_vecch')
wrap(model, '_m_step')
its = int(os.environ.get('ITERS', '10'))
torch.cuda.synchronize(); t0 = time.perf_counter()
model.train(N=its, ess_burn=10, disable=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('%d iterations, %.1f ms each (with the synchronisations of this probe)' % (its, 1e3 * dt / its))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print('  %-26s %8.2f ms / iteration   (%d calls)' % (k, 1e3 * v / its, cnt[k] / its))
