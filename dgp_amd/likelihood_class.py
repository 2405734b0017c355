"""Likelihood nodes (dgpsi likelihood_class.py): Hetero, Poisson, NegBin, ZIP, ZINB, Categorical.  They live on the host -- the 4-method plugin protocol
llik() / pllik(y, f) / prediction(m, v) / sampling(f) on numpy arrays, attributes type, name, input, output,
input_dim, exact_post_idx, rep (likelihood_class.py:30-90) -- except for the one step that is as heavy as a GP node:
the exact conditional posterior draw of the heteroskedastic Gaussian likelihood's mean latent (two n x n
factorisations per draw, likelihood_class.py:184-243), which runs on the device (Engine.post_het).
"""
import itertools

import numpy as np

from .kernel_class import TrackedInputs   # (node.input as a tracked attribute: see there)


def ghdiag(fct, mu, var, y):
    """E[exp(fct(y, f))] under f ~ N(mu, diag(var)) by the tensor-product 10-point Gauss-Hermite rule
    (functions.ghdiag, functions.py:233-241).  mu, var: (M, N) latent moments per test point; returns (M, 1)."""
    x, w = np.polynomial.hermite.hermgauss(10)
    N = mu.shape[1]
    nodes = np.array(list(itertools.product(*(x,) * N)))
    weights = np.prod(np.array(list(itertools.product(*(w,) * N))), 1)[:, None] * np.pi ** (-0.5 * N)
    fn = np.sqrt(2.0) * (np.sqrt(var[:, None]) * nodes) + mu[:, None]
    return np.sum(np.exp(np.log(weights[None, :]) + fct(y[:, None], fn)), axis=1)


class _CountLikelihood(TrackedInputs):
    """Shared state of the count likelihoods (attributes of the plugin protocol, likelihood_class.py:30-37)."""
    name = None

    def __init__(self, input_dim=None):
        self.type = 'likelihood'
        self.input = None
        self.output = None
        self.input_dim = input_dim
        self.exact_post_idx = None     # no latent with a closed-form conditional posterior: all are slice sampled
        self.rep = None


class Poisson(_CountLikelihood):
    """Poisson likelihood with log link (likelihood_class.py:8-90): y_i ~ Poisson(exp(f_i)), one feeding GP node."""
    name = 'Poisson'

    def llik(self):
        from scipy.special import gammaln
        f, y = self.input, self.output
        with np.errstate(over='ignore', invalid='ignore'):
            return np.sum(y * f - np.exp(f) - gammaln(y + 1))

    @staticmethod
    def pllik(y, f):
        from scipy.special import gammaln
        return y * f - np.exp(f) - gammaln(y + 1)

    @staticmethod
    def prediction(m, v):
        """Moments of y when f ~ N(m, v): E = exp(m + v/2), Var = E + (exp(v) - 1) exp(2m + v)  (log-normal rate)."""
        mean = np.exp(m + v / 2)
        return mean.flatten(), (mean + (np.exp(v) - 1) * np.exp(2 * m + v)).flatten()

    def sampling(self, f_sample):
        return np.random.poisson(np.exp(f_sample)).flatten()


class NegBin(_CountLikelihood):
    """Negative binomial likelihood (likelihood_class.py:245-292): mean exp(f0), dispersion exp(f1) (size 1/exp(f1)),
    two feeding GP nodes."""
    name = 'NegBin'

    @staticmethod
    def _lgamma_rise(y, size):
        """log Gamma(y + size) - log Gamma(size).  For whole counts up to 64 as sum_{j<y} log(size + j), the form the device
        log-density uses (csrc/train.hip lik_negbin): the difference of two gammaln values loses every digit once size is
        huge (size = exp(-f2) with a wild proposal), and the host path (DGPAMD_LIK_HOST=1, plugins, pllik) must take the same
        accept decisions as the device path."""
        from scipy.special import gammaln
        y, size = np.broadcast_arrays(np.asarray(y, dtype=float), np.asarray(size, dtype=float))
        out = gammaln(y + size) - gammaln(size)
        small = (y >= 0.0) & (y <= 64.0) & (y == np.floor(y))
        if small.any():
            acc = np.zeros(y.shape)
            for j in range(int(y[small].max())):
                acc = acc + np.where(j < y, np.log(size + j), 0.0)
            out = np.where(small, acc, out)
        return out

    @staticmethod
    def _logpmf(y, f1, f2):
        from scipy.special import gammaln
        with np.errstate(over='ignore', invalid='ignore', divide='ignore'):   # (wild slice-sampling proposals may overflow: they evaluate to nan / -inf and are rejected)
            size, a = np.exp(-f2), f1 + f2            # a = log(mean * dispersion); success odds exp(a)
            return NegBin._lgamma_rise(y, size) - gammaln(y + 1.0) + y * a - (y + size) * np.logaddexp(0.0, a)

    def llik(self):
        return np.sum(self._logpmf(np.asarray(self.output).flatten(), self.input[:, 0], self.input[:, 1]))

    @staticmethod
    def pllik(y, f):
        return NegBin._logpmf(y, f[:, :, [0]], f[:, :, [1]])

    @staticmethod
    def prediction(m, v):
        """E[y] = E[mu], Var[y] = Var[mu] + E[mu] + E[sigma] E[mu^2] for independent log-normal mu = exp(f0), sigma = exp(f1)."""
        e_mu = np.exp(m[:, 0] + v[:, 0] / 2)
        var = np.exp(2 * m[:, 0] + v[:, 0]) * (np.exp(v[:, 0]) - 1) + e_mu + np.exp(m[:, 1] + v[:, 1] / 2) * np.exp(2 * m[:, 0] + 2 * v[:, 0])
        return e_mu.flatten(), var.flatten()

    @staticmethod
    def sampling(f_sample):
        p, size = 1 / (1 + np.exp(f_sample[:, 0] + f_sample[:, 1])), np.exp(-f_sample[:, 1])
        return np.random.negative_binomial(size, p).flatten()


class ZIP(_CountLikelihood):
    """Zero-inflated Poisson (likelihood_class.py:470-621): with probability pi = logistic(f1) a structural zero, else
    Poisson(exp(f0)); two feeding GP nodes."""
    name = 'ZIP'

    @staticmethod
    def _logpmf(y, f_lam, f_pi):
        from scipy.special import gammaln, expit
        with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
            lam, pi = np.exp(f_lam), expit(f_pi)
            zero = np.logaddexp(np.log(pi), np.log1p(-pi) - lam)                       # structural or Poisson zero
            pos = np.log1p(-pi) - lam + y * f_lam - gammaln(y + 1.0)
            return np.where(y == 0, zero, pos)

    def llik(self):
        return np.sum(self._logpmf(np.asarray(self.output).flatten(), self.input[:, 0], self.input[:, 1]))

    @staticmethod
    def pllik(y, f):
        return ZIP._logpmf(y, f[..., [0]], f[..., [1]])

    @staticmethod
    def prediction(m, v):
        """Moments of y with a log-normal rate and the logistic-normal zero probability in its probit-style
        approximation E[pi] ~ logistic(m / sqrt(1 + pi v / 8)), Var[pi] by the delta method (clipped to p(1-p))."""
        from scipy.special import expit
        lam_mean = np.exp(m[:, 0] + 0.5 * v[:, 0])
        lam_var = (np.exp(v[:, 0]) - 1.0) * np.exp(2.0 * m[:, 0] + v[:, 0])
        den = np.maximum(1.0 + (np.pi / 8.0) * v[:, 1], 1e-12)
        p = expit(m[:, 1] / np.sqrt(den))
        p_var = np.clip((p * (1.0 - p)) ** 2 * (v[:, 1] / den), 0.0, p * (1.0 - p))
        mean = (1.0 - p) * lam_mean
        var = (1.0 - p) * lam_mean * (1.0 + p * lam_mean) + ((1.0 - p) ** 2 + p_var) * lam_var + p_var * lam_mean ** 2
        return mean.flatten(), np.maximum(var, 0.0).flatten()

    def sampling(self, f_sample):
        from scipy.special import expit
        lam, pi = np.exp(f_sample[:, 0]), expit(f_sample[:, 1])
        u = np.random.rand(f_sample.shape[0])
        return np.where(u < pi, 0, np.random.poisson(lam)).flatten()


class ZINB(_CountLikelihood):
    """Zero-inflated negative binomial (likelihood_class.py:624-812): structural zero with probability logistic(f2), else
    NegBin with mean exp(f0) and dispersion exp(f1); three feeding GP nodes."""
    name = 'ZINB'

    @staticmethod
    def _logpmf(y, f1, f2, f_pi):
        from scipy.special import expit
        with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
            nb = NegBin._logpmf(y, f1, f2)
            pi = expit(f_pi)
            return np.where(y == 0, np.logaddexp(np.log(pi), np.log1p(-pi) + nb), np.log1p(-pi) + nb)

    def llik(self):
        return np.sum(self._logpmf(np.asarray(self.output).flatten(), self.input[:, 0], self.input[:, 1], self.input[:, 2]))

    @staticmethod
    def pllik(y, f):
        return ZINB._logpmf(np.asarray(y), f[..., 0:1], f[..., 1:2], f[..., 2:3])

    @staticmethod
    def prediction(m, v):
        """Moments of y: log-normal mean mu and dispersion sigma, E[mu^2 sigma] = E[mu^2] E[sigma], zero probability as in
        ZIP; Var = E[(1-pi)(mu + mu^2 sigma)] + E[pi(1-pi)] E[mu^2] + Var[(1-pi) mu]."""
        from scipy.special import expit
        mu_mean = np.exp(m[:, 0] + 0.5 * v[:, 0])
        mu_var = (np.exp(v[:, 0]) - 1.0) * np.exp(2.0 * m[:, 0] + v[:, 0])
        mu2_mean = np.exp(2.0 * m[:, 0] + 2.0 * v[:, 0])
        mu2_sigma = mu2_mean * np.exp(m[:, 1] + 0.5 * v[:, 1])
        den = np.maximum(1.0 + (np.pi / 8.0) * v[:, 2], 1e-12)
        p = expit(m[:, 2] / np.sqrt(den))
        p_var = np.clip((p * (1.0 - p)) ** 2 * (v[:, 2] / den), 0.0, p * (1.0 - p))
        e_p1m = np.clip(p * (1.0 - p) - p_var, 0.0, p * (1.0 - p))
        var = (1.0 - p) * (mu_mean + mu2_sigma) + e_p1m * mu2_mean + ((1.0 - p) ** 2 + p_var) * mu_var + p_var * mu_mean ** 2
        return ((1.0 - p) * mu_mean).flatten(), np.maximum(var, 0.0).flatten()

    @staticmethod
    def sampling(f_sample):
        from scipy.special import expit
        size, p = np.exp(-f_sample[:, 1]), 1.0 / (1.0 + np.exp(f_sample[:, 0] + f_sample[:, 1]))
        u = np.random.rand(f_sample.shape[0])
        return np.where(u < expit(f_sample[:, 2]), 0, np.random.negative_binomial(size, p)).flatten()


class Categorical(TrackedInputs):
    """Categorical likelihood (likelihood_class.py:294-467).  Two classes: one latent, link 'logit' (default) or 'probit';
    K > 2 classes: K latents, link 'softmax' (default) or 'robustmax' (the arg-max class has probability 1 - eps, the
    others eps/(K-1)).  Outputs are class indices 0..K-1 (dgp encodes the labels).  prediction() returns class
    probabilities and their variances; for K > 2 by Monte Carlo over the latent Gaussians (1000 draws from numpy's
    global stream, antithetic for softmax)."""

    def __init__(self, num_classes=None, input_dim=None, link=None, robustmax_eps=1e-3):
        self.type = 'likelihood'
        self.name = 'Categorical'
        self.input = None
        self.output = None
        self.input_dim = input_dim
        self.exact_post_idx = None
        self.rep = None
        self.num_classes = num_classes
        self.class_encoder = None
        self.link = link
        self.robustmax_eps = robustmax_eps

    def _logp(self, y, f, axis):
        """log P(y | f); f has the classes (or the single latent) along `axis`, y holds class indices."""
        from scipy.special import log_ndtr
        with np.errstate(over='ignore', invalid='ignore'):
            if self.num_classes == 2:
                if self.link == 'logit':
                    return y * f - np.logaddexp(0, f)
                return y * log_ndtr(f) + (1 - y) * log_ndtr(-f)
            yi = np.asarray(y).astype(int)
            if self.link == 'robustmax':
                K, eps = self.num_classes, self.robustmax_eps
                hit = np.argmax(f, axis=axis) == yi
                return np.where(hit, np.log(1.0 - eps), np.log(eps / (K - 1)))
            top = np.max(f, axis=axis, keepdims=True)
            lse = np.log(np.sum(np.exp(f - top), axis=axis)) + np.squeeze(top, axis=axis)
            return np.squeeze(np.take_along_axis(f, np.expand_dims(yi, axis), axis=axis), axis=axis) - lse

    def llik(self):
        y = np.asarray(self.output)
        if self.num_classes == 2:
            return np.sum(self._logp(y, self.input, 1))
        return np.sum(self._logp(y.flatten(), self.input, 1))

    def pllik(self, y, f):
        if self.num_classes == 2:
            return self._logp(y, f, 2)
        return self._logp(np.asarray(y).reshape(-1, 1), f, 2)[:, :, None]

    def prediction(self, m, v):
        from scipy.special import expit, ndtr, owens_t
        if self.num_classes == 2:
            m, v = m.flatten(), v.flatten()
            if self.link == 'logit':       # logistic-normal mean through the probit approximation, variance by the delta method
                den = 1.0 + (np.pi / 8.0) * v
                p = expit(m / np.sqrt(den))
                pv = np.clip((p * (1.0 - p)) ** 2 * (v / den), 0.0, p * (1.0 - p))
            else:                          # probit: E[Phi(f)] = Phi(t), E[Phi(f)^2] = Phi(t) - 2 T(t, 1/sqrt(1+2v))
                t = m / np.sqrt(1.0 + v)
                p = ndtr(t)
                pv = np.maximum(p - 2.0 * owens_t(t, 1.0 / np.sqrt(1.0 + 2.0 * v)) - p * p, 0.0)
            return p.reshape(-1, 1), pv.reshape(-1, 1)
        K, S, chunk = self.num_classes, 1000, 200
        sd = np.sqrt(np.maximum(v, 0.0))
        M = m.shape[0]
        if self.link == 'robustmax':
            wins = np.zeros((M, K))
            done = 0
            while done < S:
                this = min(chunk, S - done)
                draws = m[:, None, :] + sd[:, None, :] * np.random.randn(M, this, K)
                np.add.at(wins, (np.arange(M)[:, None], np.argmax(draws, axis=2)), 1.0)
                done += this
            q = wins / S
            a, b = 1.0 - self.robustmax_eps, self.robustmax_eps / (K - 1)
            return b + (a - b) * q, (a - b) ** 2 * q * (1.0 - q)
        s1, s2 = np.zeros((M, K)), np.zeros((M, K))
        done = 0
        while done < S:
            this = min(chunk, S - done)
            half = np.random.randn(M, (this + 1) // 2, K)
            draws = m[:, None, :] + sd[:, None, :] * np.concatenate([half, -half], axis=1)[:, :this, :]
            draws -= np.max(draws, axis=2, keepdims=True)
            np.exp(draws, out=draws)
            draws /= np.sum(draws, axis=2, keepdims=True)
            s1 += draws.sum(axis=1)
            s2 += (draws * draws).sum(axis=1)
            done += this
        mean = s1 / S
        return mean, s2 / S - mean ** 2

    def sampling(self, f_sample):
        """Class probabilities at latent samples (one column for two classes: the probability of class 1)."""
        from scipy.special import expit, ndtr
        if self.num_classes == 2:
            return expit(f_sample) if self.link == 'logit' else ndtr(f_sample)
        if self.link == 'robustmax':
            K, eps = self.num_classes, self.robustmax_eps
            out = np.full_like(f_sample, eps / (K - 1), dtype=float)
            out[np.arange(f_sample.shape[0]), np.argmax(f_sample, axis=1)] = 1.0 - eps
            return out
        ex = np.exp(f_sample - np.max(f_sample, axis=1, keepdims=True))
        return ex / np.sum(ex, axis=1, keepdims=True)


class Hetero(TrackedInputs):
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
        # (a proposal that drives the log-variance latent far down makes r^2 / var overflow: the log-likelihood is then -inf, which the
        #  slice sampler rejects -- the reference's expression does the same; no warning for what is a legitimate value)
        with np.errstate(divide='ignore', over='ignore'):
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
