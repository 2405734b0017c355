"""Neighbour search at BASELINE configs[3]'s shape: ordered m-NN of n points and query-form NN of M test points, streaming
top-k kernels against the store-once kernel (DGPAMD_NN_STORE_ONCE=1), HIP-event timed."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import default_engine

eng = default_engine(0)
n, d = int(os.environ.get('N', '50000')), int(os.environ.get('D', '8'))
rng = np.random.default_rng(3)
x = eng.tensor(rng.uniform(size=(n, d)))


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    with eng.stream():
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            fn()
        e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for flag in ('0', '1'):
    os.environ['DGPAMD_NN_STORE_ONCE'] = flag
    line = ['store-once' if flag == '1' else 'streaming ']
    line.append('ordered 25-NN of %d: %.2f ms' % (n, timed(lambda: eng.nn_ordered(x, 25))))
    for M in (500, 2000, 10000, 50000):
        q = eng.tensor(rng.uniform(size=(M, d)))
        line.append('%d queries x 50: %.2f ms' % (M, timed(lambda: eng.nn_query(q, x, 50))))
    print(' | '.join(line), flush=True)
