"""dgpsi.functions with the reference's own argument lists (the njit "operator API"): a caller that did
`from dgpsi.functions import gp, link_gp, ...` swaps the import.  numpy arrays in, numpy arrays out; every function is one
call into libdgp_amd.so through the process-wide Engine (no CPU path: Engine() raises without a HIP device).

    gp          functions.py:379-394      dgpamd_gp_predict
    link_gp     functions.py:396-430      dgpamd_linkgp_predict   (R2sexp / Psexp are accepted and not needed: the
                                          D x n x n array Psexp is never materialised, IJ_sexp's terms are formed in flight)
    fmvn        functions.py:113-121      dgpamd_potrf + dgpamd_trmv_lower (the normals are drawn on the host with numpy's
                                          global generator, like the reference's np.random.randn under numba's seed)
    update_f    functions.py:203-208      dgpamd_ess_propose
"""
import numpy as np

from .ops import default_engine, raise_not_pd


def _eng(engine):
    return engine if engine is not None else default_engine()


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def gp(x, z, w1, global_w1, Rinv, Rinv_y, scale, length, nugget, name, engine=None):
    """(m, v) of a GP node at the rows of x (global inputs z / global_w1 or None), functions.py:380."""
    e = _eng(engine)
    if z is not None:
        x = np.concatenate((x, z), 1)
        w1 = np.concatenate((w1, global_w1), 1)
    n = len(w1)
    with e.stream():
        m, v = e.gp_predict(name, e.tensor(_f(x)), e.tensor(_f(w1)), _f(length), e.tensor(_f(Rinv)), n, e.tensor(_f(Rinv_y).ravel()),
                            float(np.ravel(scale)[0]), float(np.ravel(nugget)[0]))
        return e.fetch(m), e.fetch(v)


def link_gp(m, v, z, w1, global_w1, Rinv, Rinv_y, R2sexp, Psexp, scale, length, nugget, name, engine=None):
    """(m_new, v_new) of a GP node whose local inputs are normal with means m and variances v, functions.py:397."""
    e = _eng(engine)
    n = len(w1)
    with e.stream():
        mo, vo = e.linkgp_predict(name, e.tensor(_f(m)), e.tensor(_f(v)), None if z is None else e.tensor(_f(z)), e.tensor(_f(w1)),
                                  None if z is None else e.tensor(_f(global_w1)), _f(length), e.tensor(_f(Rinv)), n,
                                  e.tensor(_f(Rinv_y).ravel()), float(np.ravel(scale)[0]), float(np.ravel(nugget)[0]))
        return e.fetch(mo), e.fetch(vo)


def fmvn(cov, engine=None):
    """One draw from N(0, cov), functions.py:114: L z with L the Cholesky factor of cov."""
    e = _eng(engine)
    cov = _f(cov)
    n = len(cov)
    z = np.random.randn(n)
    Np = e.padded_dim(n)
    with e.stream():
        A = e.zeros(Np, Np)
        A[:n, :n] = e.tensor(cov)
        _, info = e.potrf(n, A)
        out = e.trmv_lower(n, A, [1.0], e.tensor(z))
        bad = int(e.fetch(info)[0])
        if bad:
            raise_not_pd(bad)   # numpy.linalg.LinAlgError, what np.linalg.cholesky raises in the reference
        return e.fetch(out)[0]


def update_f(f, nu, theta, engine=None):
    """The elliptical-slice proposal f cos(theta) + nu sin(theta), functions.py:204."""
    e = _eng(engine)
    f, nu = _f(f), _f(nu)
    with e.stream():
        out = e.ess_propose(e.tensor(f.reshape(len(f), -1)), e.tensor(nu.reshape(len(nu), -1)), np.array([float(theta)]))
        return e.fetch(out).reshape(f.shape)
