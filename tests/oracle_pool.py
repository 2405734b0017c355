"""The oracle's per-test-point predictors over a pool of host processes (test infrastructure, like oracle/ itself).

oracle.link_gp_predict walks its test points one after another -- ~1.5 s per point for the Matern-2.5 closed forms at n = 2000 -- and the
full-size GPU tests compare 16-64 points: split over spawned workers (never forked: the parent holds a HIP context; the workers import
numpy, scipy and oracle/ only) the comparisons take seconds.  Same function, same arguments, same numbers: every point is evaluated by
exactly the statements of oracle/dgp_oracle.py, the rows of the result are put back in order."""
import multiprocessing as mp
import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np


def _job(args):
    os.environ['OPENBLAS_NUM_THREADS'] = '1'
    os.environ['OMP_NUM_THREADS'] = '1'
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import dgp_oracle as O
    m, v, z, rest, kw = args
    return O.link_gp_predict(m, v, z, *rest, **kw)


def link_gp_predict(m, v, z, *rest, workers=None, **kw):
    """oracle.link_gp_predict(m, v, z, W, Wg, Rinv, Rinv_y, scale, length, nugget, name, ...) with the rows of (m, v, z) dealt to worker processes."""
    m, v = np.asarray(m, float), np.asarray(v, float)
    M = len(m)
    if workers is None:
        workers = max(1, min(16, (os.cpu_count() or 2) // 2, M))
    if workers <= 1 or M < 4:
        return _job((m, v, z, rest, kw))
    cuts = np.linspace(0, M, workers + 1).astype(int)
    jobs = [(m[a:b], v[a:b], None if z is None else np.asarray(z)[a:b], rest, kw) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    with ProcessPoolExecutor(max_workers=len(jobs), mp_context=mp.get_context('spawn')) as ex:
        res = list(ex.map(_job, jobs))
    return np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res])
