"""dgpsi.vecchia with the reference's own argument lists (the njit "operator API"): a caller that did
`from dgpsi.vecchia import nn, vecchia_llik, ...` swaps the import.  numpy arrays in, numpy arrays out; every function is
one call into libdgp_amd.so through the process-wide Engine (no CPU path).

    get_pred_nn       vecchia.py:20-40     dgpamd_nn_query    (exact search in (distance, index) order; `method`, `size`,
    nn                vecchia.py:61-109    dgpamd_nn_ordered   `efSearch`, `n_jobs` select faiss / sklearn back ends there
                                                               and are accepted and ignored here)
    forward_solve_sp  vecchia.py:112-120   dgpamd_vecchia_spsolve
    vecchia_llik      vecchia.py:165-180   dgpamd_vecchia_llik
    vecchia_nllik     vecchia.py:183-242   dgpamd_vecchia_nllik + the closing scale / replicate algebra (nllik_close)
    L_matrix          vecchia.py:410-424   dgpamd_vecchia_lmatrix
    gp_vecch          vecchia.py:636-654   dgpamd_vecchia_gp
    link_gp_vecch     vecchia.py:759-796   dgpamd_vecchia_linkgp
"""
import numpy as np

from .ops import default_engine


def _eng(engine):
    return engine if engine is not None else default_engine()


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _nn_dev(e, NNarray):
    import torch
    return e.tensor(np.ascontiguousarray(NNarray, dtype=np.int64), dtype=torch.int64)


def get_pred_nn(query, x, m=50, method='exact', size=40, efSearch=100, n_jobs=-1, engine=None):
    """The m nearest rows of x for every row of query, nearest first (all of them, cyclically, when m >= len(x))."""
    e = _eng(engine)
    with e.stream():
        return e.fetch(e.nn_query(e.tensor(_f(query)), e.tensor(_f(x)), int(m)))


def nn(x, m, method='exact', size=40, efSearch=100, n_jobs=-1, engine=None):
    """NNarray (n x (m+1), int64): row i = i, then its <= m nearest earlier points by index descending, -1 padding."""
    e = _eng(engine)
    with e.stream():
        return e.fetch(e.nn_ordered(e.tensor(_f(x)), int(m)))


def forward_solve_sp(L, NNarray, b, engine=None):
    """x with x_i = (b_i - sum_{j >= 1} L[i, j] x[NNarray[i, j]]) / L[i, 0]."""
    e = _eng(engine)
    with e.stream():
        return e.fetch(e.vecchia_spsolve(e.tensor(_f(L)), _nn_dev(e, NNarray), 1.0, e.tensor(_f(b).ravel())))


def vecchia_llik(X, y, NNarray, scale, length, nugget, nugget_diag, name, engine=None):
    """-0.5 (logdet + quad / scale) summed over the rows' conditionals; a (1,) array like the reference's."""
    e = _eng(engine)
    with e.stream():
        o = e.fetch(e.vecchia_llik(name, e.tensor(_f(X)), e.tensor(_f(y).reshape(len(X), -1)[:, 0].copy()), _nn_dev(e, NNarray), _f(length),
                                   float(np.ravel(nugget)[0]), e.tensor(_f(nugget_diag))))
    return np.atleast_1d(-0.5 * (o[1] + o[0] / float(np.ravel(scale)[0])))


def nllik_close(o, P, n, origin_n, rr, scale, nugget, scale_est, nugget_est):
    """The closing algebra of vecchia_nllik (vecchia.py:224-241) on the reduced device sums o = [quad, logdet, dquad (P),
    dlogdet (P)]: profile out the scale or not; with replicates (n sites standing for origin_n observations, residual sum
    rr) the extra nugget terms.  Returns (nllik, gradient wrt the log-parameters, scale)."""
    quad, logdet, dquad, dlogdet = o[0], o[1], o[2:2 + P].copy(), o[2 + P:].copy()
    reps = n != origin_n
    if scale_est:
        scale = (quad + rr / nugget) / origin_n if reps else quad / n
        nll = 0.5 * (logdet + (origin_n if reps else n) * np.log(scale))
        g = 0.5 * (dlogdet - dquad / scale)
        if reps and nugget_est:
            nll += 0.5 * (origin_n - n) * np.log(nugget)
            g[-1] += 0.5 * (-rr / (scale * nugget) + (origin_n - n))
    else:
        nll = 0.5 * (logdet + quad / scale)
        g = 0.5 * (dlogdet - dquad / scale)
        if reps and nugget_est:
            nll += 0.5 * (rr / (nugget * scale) + (origin_n - n) * np.log(nugget))
            g[-1] += 0.5 * (-rr / (scale * nugget) + (origin_n - n))
    return nll, g, scale


def vecchia_nllik(X, y, NNarray, scale, length, nugget, nugget_diag, name, scale_est, nugget_est, origin_n, rr, engine=None):
    """(nllik (1,), gradient (p,), scale (1,)): the M-step objective of a Vecchia GP node and its derivatives wrt the log
    lengthscale(s) and, with nugget_est, the log nugget."""
    e = _eng(engine)
    n = len(X)
    nug = float(np.ravel(nugget)[0])
    with e.stream():
        o, P = e.vecchia_nllik(name, e.tensor(_f(X)), e.tensor(_f(y).reshape(n, -1)[:, 0].copy()), _nn_dev(e, NNarray), _f(length), nug,
                               e.tensor(_f(nugget_diag)), bool(nugget_est))
        o = e.fetch(o)
    nll, g, sc = nllik_close(o, P, n, int(origin_n), float(rr), float(np.ravel(scale)[0]), nug, bool(scale_est), bool(nugget_est))
    return np.atleast_1d(nll), g, np.array([sc])


def L_matrix(X, NNarray, length, nugget, name, engine=None):
    """Row i: the last row of the inverse Cholesky factor of the conditioning block of point i, self first."""
    e = _eng(engine)
    with e.stream():
        return e.fetch(e.vecchia_lmatrix(name, e.tensor(_f(X)), _nn_dev(e, NNarray), _f(length), float(np.ravel(nugget)[0])))


def gp_vecch(x, w, NNarray, y, scale, length, nugget, nugget_diag, name, engine=None):
    """(m, v) of a GP node at the rows of x, each conditioned on its own neighbours NNarray[i] among the rows of w."""
    e = _eng(engine)
    with e.stream():
        m, v = e.vecchia_gp(name, e.tensor(_f(x)), e.tensor(_f(w)), _nn_dev(e, NNarray), e.tensor(_f(y).reshape(len(w), -1)[:, 0].copy()),
                            float(np.ravel(scale)[0]), _f(length), float(np.ravel(nugget)[0]), e.tensor(_f(nugget_diag)))
        return e.fetch(m), e.fetch(v)


def link_gp_vecch(m, v, z, w1, global_w1, NNarray, y, scale, length, nugget, nugget_diag, name, engine=None):
    """(m_new, v_new) for normally distributed local inputs (means m, variances v), neighbour sets NNarray."""
    e = _eng(engine)
    with e.stream():
        mo, vo = e.vecchia_linkgp(name, e.tensor(_f(m)), e.tensor(_f(v)), None if z is None else e.tensor(_f(z)), e.tensor(_f(w1)),
                                  None if z is None else e.tensor(_f(global_w1)), _nn_dev(e, NNarray),
                                  e.tensor(_f(y).reshape(len(w1), -1)[:, 0].copy()), float(np.ravel(scale)[0]), _f(length),
                                  float(np.ravel(nugget)[0]), e.tensor(_f(nugget_diag)))
        return e.fetch(mo), e.fetch(vo)
