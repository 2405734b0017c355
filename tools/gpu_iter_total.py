"""Wall time per SI iteration (no synchronisation inside the loop, as bench.py) under the factorisation modes and with /
without the device-queued ESS loop; every configuration starts from a fresh model (the iteration gets longer as training
proceeds, so the configurations must see the same iterations)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model

N = int(os.environ.get('ITS', '30'))
for mode, queued in ((1, True), (1, False), (0, False), (0, True), (2, True), (1, True)):
    model, X, Y = build_model(2000, 5, 100, 0)
    imp, eng = model.imp, model.engine
    eng.set_potrf_mode(mode)
    imp.queued = queued
    for _ in range(3):
        imp.sample(burnin=10); model._m_step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(N):
        imp.sample(burnin=10); model._m_step()
    torch.cuda.synchronize()
    print('potrf mode %d, queued ESS %-5s: %.1f ms per iteration over %d iterations' % (mode, queued, 1e3 * (time.perf_counter() - t) / N, N), imp.stats)
