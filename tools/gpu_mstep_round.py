"""Time the phases of one batched M-step round (B=6, n=2000)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model

model, X, Y = build_model(2000, 5, 100, 0)
e = model.engine
nodes = [nd for layer in model.all_layer for nd in layer]
for nd in nodes:
    nd._stage()
n, B = 2000, len(nodes)
Np = e.padded_dim(n)
A = e.empty(B, Np, Np); Ainv = e.empty(B, Np, Np)
work = e.potrf_workspace(n, B)


def phase(f, reps=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps


def km():
    for b, nd in enumerate(nodes):
        s = nd._staged
        e.kmatrix(nd.name, s['Xl'], None, s['Xg'], nd.length, nd.nugget[0], W=s['W'], out=A[b], full=False, Y=s['y'])


def gr():
    for b, nd in enumerate(nodes):
        s = nd._staged
        e.grad_reduce(nd.name, s['Xl'], None, s['Xg'], nd.length, nd.nugget[0], nd.nugget_est, Ainv[b], W=s['W'])


print('kmatrix x%d: %.2f ms' % (B, phase(km)))
print('potrf B=%d: %.2f ms' % (B, phase(lambda: (km(), e.potrf(n, A, batch=B, work=work))) - phase(km)))
km(); e.potrf(n, A, batch=B, work=work)
def pi():
    e.potri(n, A, Ainv, 1, work, batch=B)
print('potri B=%d: %.2f ms (on already inverted input: timing only)' % (B, phase(pi)))
print('grad_reduce x%d: %.2f ms' % (B, phase(gr)))
for b in (1, 2, 4):
    print('potri B=%d: %.2f ms' % (b, phase(lambda: e.potri(n, A, Ainv, 1, work, batch=b))))
