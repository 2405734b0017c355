"""Small host utilities that keep dgpsi's names (utils.py:51-66, 203-269)."""
import numpy as np

_threads = [8]


def nb_seed(value):
    """dgpsi.nb_seed seeds numba's RNG (utils.py:51-55); here the global numpy RNG, which only the
    one-off initialisers use -- the sampler's streams are seeded through dgp(..., seed=) / emulator(..., seed=)."""
    np.random.seed(value)


def set_thread(value):
    """Number of GP nodes optimised concurrently in the M-step (one HIP stream each).
    The reference's set_thread sizes numba's thread pool (utils.py:63-66)."""
    _threads[0] = max(1, int(value))


def get_thread():
    return _threads[0]


class NystromKPCA:
    """Nystrom approximation of sigmoid kernel PCA used to warm-start hidden layers when n >= 500
    (one-off host initialisation; restates utils.py:203-269)."""

    def __init__(self, n_components, m=200):
        self.m, self.n_components = m, n_components
        self.basis_inds = None

    @staticmethod
    def _pinv(K, sqrt=False):
        U, S, V = np.linalg.svd(K)
        S = np.maximum(S, 1e-12)
        return (U / (np.sqrt(S) if sqrt else S)) @ V

    def fit_transform(self, X):
        from sklearn.metrics.pairwise import pairwise_kernels
        n = X.shape[0]
        m = self.m = min(n, self.m)
        self.basis_inds = np.random.permutation(n)[:m]
        Knm = pairwise_kernels(X, X[self.basis_inds], metric='sigmoid', filter_params=True)
        Kmm = Knm[self.basis_inds]
        mean_n = Knm.sum(0) / n
        m0 = self._pinv(Kmm) @ mean_n[:, None]
        M3 = mean_n @ m0
        Knm_c = Knm - mean_n[None, :] - Knm @ m0 + M3
        Kmm_c = Kmm - mean_n[None, :] - mean_n[:, None] + M3
        Kis = self._pinv(Kmm_c, sqrt=True)
        _, U = np.linalg.eigh(Kis @ Knm_c.T @ Knm_c @ Kis / n)
        scores = Knm_c @ (Kis @ U[:, ::-1][:, :self.n_components])
        flip = (scores.min(0) + scores.max(0)) / 2 < 0
        return scores * (1 - 2 * flip)[None, :]
