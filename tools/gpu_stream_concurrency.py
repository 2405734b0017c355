"""How well do independent n=2000 factorisations on separate HIP streams overlap? (M-step concurrency)"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine

n = 2000
rng = np.random.default_rng(0)
main = Engine(0)
Np = main.padded_dim(n)
Xh, yh = rng.uniform(size=(n, 5)), rng.normal(size=n)


def make(eng):
    with eng.stream():
        X, y = eng.tensor(Xh), eng.tensor(yh)
        A = eng.empty(Np, Np)
        Ainv = eng.empty(Np, Np)
        return X, y, A, Ainv


def work(eng, bufs, reps, with_sync):
    X, y, A, Ainv = bufs
    with eng.stream():
        w = eng.potrf_workspace(n, 1)
        for _ in range(reps):
            eng.kmatrix('matern2.5', X, None, None, [1.0], 1e-6, out=A, full=False, Y=y)
            ld, info = eng.potrf(n, A, work=w)
            eng.potri(n, A, Ainv, 1, w)
            if with_sync:
                ld.cpu()
        eng._torch_stream.synchronize()


for T in (1, 2, 4, 6, 8):
    engines = [Engine(0, torch.cuda.Stream()) for _ in range(T)]
    bufs = [make(e) for e in engines]
    for with_sync in (False, True):
        for e, b in zip(engines, bufs):
            work(e, b, 2, with_sync)
        torch.cuda.synchronize()
        reps = 10
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(e, b, reps, with_sync)) for e, b in zip(engines, bufs)]
        [t.start() for t in th]
        [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('%d streams, sync-per-eval=%s: %.2f ms per (potrf+potri) per stream, aggregate %.0f evals/s' % (T, with_sync, 1e3 * dt / reps, T * reps / dt))
