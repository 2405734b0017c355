"""potrf / potrf_inv at the bench size by batch: median and minimum of `reps` HIP-event times (one library per process: DGPAMD_LIB picks another build).
usage: python tools/gpu_potrf_table.py [n] [reps] [batches...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
batches = [int(v) for v in sys.argv[3:]] or [1, 2, 3, 4, 6, 10, 12]
e = Engine(0)
rng = np.random.default_rng(0)
Np = e.padded_dim(n)
ev = (e.event(), e.event())
print('lib %s  split_min %s  n=%d  median / min ms over %d' % (os.path.basename(os.environ.get('DGPAMD_LIB', 'libdgp_amd.so')), os.environ.get('DGPAMD_MEGA_SPLIT_MIN', '-'), n, reps))
for B in batches:
    X, y = e.tensor(rng.uniform(size=(B, n, 5))), e.tensor(rng.normal(size=n))
    A, T, S = e.empty(B, Np, Np), e.empty(B, Np, Np), e.empty(B, Np, Np)
    work = e.potrf_workspace(n, B)
    out = []
    for inv in (0, 1):
        ts = []
        for rep in range(reps + 1):
            e.kmatrix('matern2.5', X, None, None, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
            e.record(ev[0])
            if inv:
                e.potrf_inv(n, A, T, S, batch=B, work=work)
            else:
                e.potrf(n, A, batch=B, work=work)
            e.record(ev[1])
            torch.cuda.synchronize()
            if rep:
                ts.append(e.elapsed_ms(*ev))
        out.append('%6.3f / %6.3f' % (float(np.median(ts)), min(ts)))
    print(' B=%2d  potrf %s   potrf_inv %s' % (B, out[0], out[1]), flush=True)
    del A, T, S
