"""Host time of the pieces between the I-step's last kernel and the M-step's first round (and back), bench shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import build_model
from dgp_amd import mstep
model, X, Y = build_model(2000, 5, 100, 0)
imp = model.imp
for _ in range(3):
    imp.sample(burnin=10); model._m_step()
acc = {}
def timed(obj, name, label):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, w)
timed(imp, '_attach', 'attach'); timed(imp, '_detach', 'detach'); timed(imp, '_prior_draws_ahead', 'prior draws (enqueue)')
timed(imp, 'stage_for_mstep', 'stage_for_mstep'); timed(imp, '_layer_factors', '  of which layer factors (incl. sync)')
orig_ml = mstep.maximise_lockstep
orig_min = mstep.minimize_lockstep
def min_w(problems, evaluate):
    t = time.perf_counter(); r = orig_min(problems, evaluate); acc['minimize_lockstep (rounds)'] = acc.get('minimize_lockstep (rounds)', 0.0) + time.perf_counter() - t; return r
mstep.minimize_lockstep = min_w
def ml_w(*a, **k):
    t = time.perf_counter(); r = orig_ml(*a, **k); acc['maximise_lockstep total'] = acc.get('maximise_lockstep total', 0.0) + time.perf_counter() - t; return r
mstep.maximise_lockstep = ml_w
import dgp_amd.dgp as D
N = 20
ti = tm = 0.0
for _ in range(N):
    t0 = time.perf_counter(); imp.sample(burnin=10); t1 = time.perf_counter(); model._m_step(); t2 = time.perf_counter()
    ti += t1 - t0; tm += t2 - t1
print('per iteration: sample() %.2f ms, _m_step() %.2f ms' % (1e3 * ti / N, 1e3 * tm / N))
for k, v in acc.items():
    print('  %-40s %.3f ms' % (k, 1e3 * v / N))
