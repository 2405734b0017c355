"""Host-side wall-time breakdown of one SI iteration at the bench shapes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model
from dgp_amd import kernel_class

model, X, Y = build_model(2000, 5, 100, 0)
imp = model.imp
imp.batch = int(os.environ.get("ESS_BATCH", "12"))
imp.batch_next = int(os.environ.get("ESS_BATCH_NEXT", "0")) or None
for _ in range(2):
    imp.sample(burnin=10); model._m_step()
T = dict(i=0.0, m=0.0, prior=0.0, upper=0.0, attach=0.0, detach=0.0)
orig_prior, orig_upper, orig_att, orig_det = imp._prior_draw, imp._upper_loglik, imp._attach, imp._detach
def timed(name, f):
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); T[name] += time.perf_counter() - t; return r
    return g
imp._prior_draw = timed('prior', orig_prior); imp._upper_loglik = timed('upper', orig_upper)
imp._attach = timed('attach', orig_att); imp._detach = timed('detach', orig_det)
calls = [0]
orig = kernel_class.kernel.llik
tl = [0.0]
def ll(self, x):
    calls[0] += 1
    return orig(self, x)
kernel_class.kernel.llik = ll
N = 8
for _ in range(N):
    torch.cuda.synchronize(); t = time.perf_counter(); imp.sample(burnin=10); torch.cuda.synchronize(); T['i'] += time.perf_counter() - t
    t = time.perf_counter(); model._m_step(); torch.cuda.synchronize(); T['m'] += time.perf_counter() - t
print('per SI iteration (ms): I-step %.1f [prior draws %.1f, upper logliks %.1f, attach %.1f, detach %.1f] | M-step %.1f (%.1f llik calls, %.2f ms each if serial)'
      % (1e3 * T['i'] / N, 1e3 * T['prior'] / N, 1e3 * T['upper'] / N, 1e3 * T['attach'] / N, 1e3 * T['detach'] / N, 1e3 * T['m'] / N, calls[0] / N, 1e3 * T['m'] / max(1, calls[0])))
st = imp.stats
print('proposals/update %.2f batches/update %.2f' % (st['proposals'] / st['updates'], st['batches'] / st['updates']))
# single-stream llik timing
nd = model.all_layer[0][0]
x = nd.log_t()
for _ in range(3): orig(nd, x)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): orig(nd, x)
torch.cuda.synchronize(); print('one llik (layer-1 node, serial): %.2f ms' % (1e2 * (time.perf_counter() - t)))
nd = model.all_layer[1][0]
x = nd.log_t()
for _ in range(3): orig(nd, x)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): orig(nd, x)
torch.cuda.synchronize(); print('one llik (layer-2 node, serial): %.2f ms' % (1e2 * (time.perf_counter() - t)))
