"""Identity stub of the `numba` names dgpsi imports (TEST INFRASTRUCTURE, this container only).

The reference's @njit bodies are plain numpy-Python; with these stand-ins they run
unmodified (slowly) so that oracle/gen_golden.py can record golden vectors.
Never shipped to / used on the GPU box; never imported by the product package.
"""
import numpy as _np


def _identity_decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(fn):
        return fn
    return wrap


njit = _identity_decorator
jit = _identity_decorator
prange = range


def vectorize(*args, **kwargs):
    def wrap(fn):
        return _np.vectorize(fn, otypes=[float])
    return wrap


def float64(*args, **kwargs):
    return None


class _Config:
    NUMBA_NUM_THREADS = 8
    THREADING_LAYER = 'default'


config = _Config()
_threads = [8]


def set_num_threads(n):
    _threads[0] = int(n)


def get_num_threads():
    return _threads[0]
