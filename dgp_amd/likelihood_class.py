"""Likelihood nodes (dgpsi likelihood_class.py).  They live on the host -- the 4-method plugin protocol
llik() / pllik(y, f) / prediction(m, v) / sampling(f) on numpy arrays, attributes type, name, input, output,
input_dim, exact_post_idx, rep (likelihood_class.py:30-90) -- except for the one step that is as heavy as a GP node:
the exact conditional posterior draw of the heteroskedastic Gaussian likelihood's mean latent (two n x n
factorisations per draw, likelihood_class.py:184-243), which runs on the device (Engine.post_het).
"""
import itertools

import numpy as np


def ghdiag(fct, mu, var, y):
    """E[exp(fct(y, f))] under f ~ N(mu, diag(var)) by the tensor-product 10-point Gauss-Hermite rule
    (functions.ghdiag, functions.py:233-241).  mu, var: (M, N) latent moments per test point; returns (M, 1)."""
    x, w = np.polynomial.hermite.hermgauss(10)
    N = mu.shape[1]
    nodes = np.array(list(itertools.product(*(x,) * N)))
    weights = np.prod(np.array(list(itertools.product(*(w,) * N))), 1)[:, None] * np.pi ** (-0.5 * N)
    fn = np.sqrt(2.0) * (np.sqrt(var[:, None]) * nodes) + mu[:, None]
    return np.sum(np.exp(np.log(weights[None, :]) + fct(y[:, None], fn)), axis=1)


class Hetero:
    """Heteroskedastic Gaussian likelihood (likelihood_class.py:94-243): y_i ~ N(f_i0, exp(f_i1)) with the two latents
    coming from the two GP nodes `input_dim` of the feeding layer.  Final layer only."""

    def __init__(self, input_dim=None):
        self.type = 'likelihood'
        self.name = 'Hetero'
        self.input = None
        self.output = None
        self.input_dim = input_dim
        self.exact_post_idx = np.array([0])   # the mean latent has an exact conditional posterior
        self.rep = None

    def llik(self):
        """Log-likelihood of the data at the current latents (likelihood_class.py:108-113)."""
        mu, log_var = self.input[:, 0], self.input[:, 1]
        r2 = (np.asarray(self.output).flatten() - mu) ** 2
        with np.errstate(divide='ignore'):
            return np.sum(-0.5 * (np.log(2 * np.pi) + log_var + np.exp(np.log(r2) - log_var)))

    @staticmethod
    def pllik(y, f):
        """Pointwise log-likelihood for quadrature nodes f (..., 2) (likelihood_class.py:115-121)."""
        mu, var = f[:, :, [0]], np.exp(f[:, :, [1]])
        return -0.5 * (np.log(2 * np.pi * var) + (y - mu) ** 2 / var)

    @staticmethod
    def prediction(m, v):
        """Predictive mean / variance of y from the latents' moments (likelihood_class.py:123-127)."""
        return m[:, 0].flatten(), (np.exp(m[:, 1] + v[:, 1] / 2) + v[:, 0]).flatten()

    @staticmethod
    def sampling(f_sample):
        """y ~ N(f0, exp(f1)) (likelihood_class.py:129-132)."""
        return np.random.normal(f_sample[:, 0], np.sqrt(np.exp(f_sample[:, 1]))).flatten()

    def posterior_terms(self, n_sites):
        """(gamma_eff, y_eff) of the exact-posterior draw of the mean latent given the log-variance latent in
        self.input[:, 1]: without replicates (Gamma, y) (post_het1); with replicates the per-site precision-weighted
        sums 1 / (M' Gamma^-1 M) and their product with M' Gamma^-1 y (post_het2, likelihood_class.py:214-230)."""
        Gamma = np.exp(self.input[:, 1])
        y = np.asarray(self.output, dtype=float).flatten()
        if self.rep is None:
            return Gamma, y
        Gi = 1.0 / Gamma
        MGy = np.bincount(self.rep, weights=Gi * y, minlength=n_sites)
        iMGM = 1.0 / np.bincount(self.rep, weights=Gi, minlength=n_sites)
        return iMGM, iMGM * MGy

    def posterior(self, idx, v, sd=None, engine=None):
        """Draw of latent `idx` (only 0, the mean) from its exact conditional posterior given the covariance v (n x n
        numpy, = scale * k_matrix() of the feeding GP node) -- likelihood_class.py:134-151, evaluated on the device.
        sd: (n, 2) standard normals (drawn from numpy's global stream like the reference if None)."""
        if int(np.asarray(idx).reshape(-1)[0]) != 0:
            return None
        from .ops import default_engine
        e = engine if engine is not None else default_engine()
        n = v.shape[0]
        if sd is None:
            sd = np.random.randn(n, 2)
        g, y = self.posterior_terms(n)
        return e.post_het(e.tensor(v), 1.0, e.tensor(g), e.tensor(y), e.tensor(sd)).cpu().numpy()
