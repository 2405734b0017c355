"""K assembly against the HBM roofline: time vs tile count at (nearly) fixed bytes, rotating outputs so that
no launch finds its lines in the 256-MB Infinity Cache, HIP events around every launch (dgpamd_prof), and a write-only
torch fill_ of the same buffers as the box's ceiling.  usage (GPU box): python tools/gpu_kmatrix_roofline.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine

e = Engine(0)
rng = np.random.default_rng(0)
print('%-10s %6s %3s %6s %3s %4s | %8s %9s %8s | %9s %8s' % ('kernel', 'n', 'D', 'mode', 'B', 'bufs', 'us', 'GB/s(alg)', 'of 8TB/s', 'fill_ GB/s', 'of 8TB/s'))
shapes = tuple(tuple(int(v) if i != 2 else bool(int(v)) for i, v in enumerate(t.split(','))) for t in os.environ['SHAPES'].split()) if os.environ.get('SHAPES') else ((5000, 10, True, 1, 3), (5000, 10, True, 4, 2), (8192, 10, True, 1, 3), (10000, 10, True, 1, 3), (16384, 10, True, 1, 2),
          (2000, 5, False, 12, 3), (2000, 10, False, 6, 3), (5000, 10, False, 10, 2))
for name in ('sexp', 'matern2.5'):
    for (n, D, full, B, nbuf) in shapes:
        X = e.tensor(rng.uniform(size=(B, n, D)))
        ld = n if full else e.padded_dim(n)
        pad = int(os.environ.get('LDPAD', '0')) if full else 0   # (full mode: rows of n + pad doubles, the matrix a view of them)
        if pad and B == 1:
            outs = [e.empty(n, n + pad)[:, :n] for _ in range(nbuf)]
        else:
            outs = [e.empty(B, ld, ld) if B > 1 else e.empty(ld, ld) for _ in range(nbuf)]
        length = np.full(D, 0.9)
        for o in outs:
            e.kmatrix(name, X if B > 1 else X[0], None, None, length, 1e-6, out=o, full=full, batch=B)
        torch.cuda.synchronize()
        e.prof_enable('kmatrix')
        reps = 24
        for r in range(reps):
            e.kmatrix(name, X if B > 1 else X[0], None, None, length, 1e-6, out=outs[r % nbuf], full=full, batch=B)
        k, ms, w = e.prof_collect()
        us = 1e3 * ms / k
        nbytes = B * ((8.0 * n * n) if full else (4.0 * ld * ld)) + 8.0 * B * n * D
        # the same buffers filled by torch (write-only: the ceiling of this box for this footprint)
        for o in outs:
            o.fill_(1.5)
        fb = 8.0 * outs[0].numel()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with e.stream():
            st = torch.cuda.current_stream()
            e0.record(st)
            for r in range(reps):
                outs[r % nbuf].fill_(1.5)
            e1.record(st)
        torch.cuda.synchronize()
        fms = e0.elapsed_time(e1) / reps
        print('%-10s %6d %3d %6s %3d %4d | %8.1f %9.0f %8.3f | %9.0f %8.3f' % (name, n, D, 'full' if full else 'lower', B, nbuf, us, nbytes / us / 1e3,
                                                                         nbytes / us / 1e3 / 8000, fb / fms / 1e6, fb / fms / 1e6 / 8000), flush=True)
        del outs, X
