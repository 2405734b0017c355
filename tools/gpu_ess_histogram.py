"""Distribution of the number of ESS proposals per layer update on the bench workload (how far the shrinking bracket
has to go before a proposal is accepted) -- the input for choosing the speculative batch sizes."""
import sys, os, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dgp_amd import ops

model, X, Y = bench.build_model(2000, 5, 100, 0)
hist = collections.Counter()
orig = ops._EssPlan.run

def run(self, *a, **k):
    out = orig(self, *a, **k)
    if out[0] == 0:
        run.acc += out[2]
        hist[run.acc] += 1
        run.acc = 0
    else:
        run.acc += out[2]
    return out
run.acc = 0
ops._EssPlan.run = run
for it in range(12):
    model.imp.sample(burnin=10)
    model._m_step()
    if it == 1:
        hist.clear()
tot = sum(hist.values())
print('updates', tot, 'mean proposals %.2f' % (sum(k * v for k, v in hist.items()) / tot))
cum = 0
for k in sorted(hist):
    cum += hist[k]
    print('%3d proposals: %4d  cumulative %.3f' % (k, hist[k], cum / tot))
