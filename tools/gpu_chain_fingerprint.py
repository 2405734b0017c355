"""Fingerprint of the bench model's SI chain: per iteration a hash of the hidden layer's latents after the I-step and of the hyper-parameters after the M-step,
the proposals and optimiser rounds -- to check that settings which must not change the chain (speculative batch sizes of the device queue, DGPAMD_ESS_BATCH)
do not.   usage: python tools/gpu_chain_fingerprint.py [iterations]"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dgp_amd import mstep as mstep_mod

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
model, X, Y = bench.build_model(2000, 5, 100, 0)
rounds = []
orig = mstep_mod.minimize_lockstep
mstep_mod.minimize_lockstep = lambda *a, **k: (rounds.append(orig(*a, **k)) or rounds[-1])
h = lambda a: hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
print('DGPAMD_ESS_BATCH =', os.environ.get('DGPAMD_ESS_BATCH'))
for it in range(N):
    p0 = model.imp.stats['proposals']
    model.imp.sample(burnin=10)
    F = np.stack([nd.output[:, 0] for nd in model.all_layer[0]], 1)
    model._m_step()
    hy = np.concatenate([np.concatenate((nd.scale, nd.length, nd.nugget)) for layer in model.all_layer for nd in layer])
    print('it %2d  latents %s  hyper %s  proposals %3d  rounds %2d  queued %d' % (it, h(F), h(hy), model.imp.stats['proposals'] - p0, rounds[-1], model.imp.queued_calls))
