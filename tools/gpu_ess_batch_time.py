"""Wall time per SI iteration at the bench shape for different speculative batch sizes (first, next); every configuration
starts from a fresh model and sees the same iterations (the sampler's trajectory does not depend on the batch sizes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model

N = int(os.environ.get('ITS', '40'))
for b1, b2 in ((12, 4), (10, 4), (8, 4), (10, 6), (9, 5), (12, 4), (10, 4)):
    model, X, Y = build_model(2000, 5, 100, 0)
    imp = model.imp
    imp.batch, imp.batch_next = b1, b2
    for _ in range(3):
        imp.sample(burnin=10); model._m_step()
    torch.cuda.synchronize(); t = time.perf_counter()
    ti = 0.0
    for _ in range(N):
        t0 = time.perf_counter(); imp.sample(burnin=10); torch.cuda.synchronize(); ti += time.perf_counter() - t0
        model._m_step()
    torch.cuda.synchronize()
    print('batch %2d / %d: %.2f ms per iteration (I-step %.2f) over %d iterations' % (b1, b2, 1e3 * (time.perf_counter() - t) / N, 1e3 * ti / N, N),
          {k: imp.stats[k] for k in ('proposals', 'batches', 'updates')})
