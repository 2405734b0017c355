"""Single-GP emulator -- mirror of dgpsi.gp (gp.py:12-453; training, export, mean/variance prediction)."""
import copy

import numpy as np
from .kernel_class import bind_private, peek

from .ops import default_engine


class gp:
    """Args as dgpsi.gp (gp.py:26): X (n x d), Y (n x 1), kernel, check_rep, vecchia, m, ord_fun; plus `device`."""

    def __init__(self, X, Y, kernel, check_rep=True, vecchia=False, m=25, ord_fun=None, device=None):
        if Y.ndim == 1 or X.ndim == 1:
            raise Exception('The input and output data have to be numpy 2d-arrays.')
        self.check_rep = check_rep
        self._set_data(X, Y)
        self.kernel = kernel
        self.kernel.engine = default_engine(device)
        self._finish_init(vecchia, m, ord_fun)

    def _set_data(self, X, Y):
        """Deduplicate the design (gp.py:32-42,152-170): site means, weights 1/count and the pooled residual."""
        self.indices = None
        self.X, self.Y = X, Y
        if self.check_rep:
            X0, inv = np.unique(X, return_inverse=True, axis=0)
            if len(X0) != len(X):
                inv = np.asarray(inv).reshape(-1)
                G = inv.max() + 1
                self.X, self.indices = X0, inv
                self.W_diag = 1.0 / np.bincount(inv, minlength=G)
                self.Y = (np.bincount(inv, weights=Y.flatten(), minlength=G) * self.W_diag).reshape(-1, 1)
                res = Y - self.Y[inv, :]
                self.sum_residual = (res.T @ res).flatten()

    def _finish_init(self, vecchia, m, ord_fun):
        self.vecch = vecchia
        self.n_data = self.X.shape[0]
        self.m = min(m, self.n_data - 1)
        self.ord_fun = ord_fun
        self.initialize()
        if self.vecch:
            self.kernel.ord_nn()
        else:
            self.kernel.compute_stats()

    def initialize(self):
        """Hand the data to the node (gp.py:80-113)."""
        k = self.kernel
        if k.input_dim is None:
            k.input_dim = np.arange(self.X.shape[1])
            bind_private(k, 'input', self.X.copy())
        else:
            bind_private(k, 'input', self.X[:, k.input_dim])
        if self.indices is not None:
            k.rep, k.W_diag, k.sum_residual = self.indices, self.W_diag, self.sum_residual
        if k.connect is not None:
            if len(np.intersect1d(k.connect, k.input_dim)) != 0:
                raise Exception('The local input and global input should not have any overlap. Change input_dim or '
                                'connect so they do not have any common indices.')
            bind_private(k, 'global_input', self.X[:, k.connect])
        k.output = self.Y.copy()
        k.D = peek(k, 'input').shape[1] + (0 if k.connect is None else len(k.connect))
        k.para_path = np.atleast_2d(np.concatenate((k.scale, k.length, k.nugget)))
        k.vecch, k.m = self.vecch, self.m
        if self.ord_fun is not None:
            k.ord_fun = self.ord_fun
        if k.prior_name == 'ref':
            k.prior_coef = np.concatenate((k.prior_coef, 1 / self.n_data ** (1 / k.D) * (k.prior_coef + k.D)))
            k.compute_cl()
        k.target = 'gp'

    def to_vecchia(self, m=25, ord_fun=None):
        if self.vecch:
            raise Exception('The GP emulator is already in Vecchia mode.')
        self.vecch, self.m, self.ord_fun = True, min(m, self.n_data - 1), ord_fun
        self.kernel.vecch, self.kernel.m, self.kernel.ord_fun = True, self.m, ord_fun
        self.kernel.ord_nn()

    def remove_vecchia(self):
        if not self.vecch:
            raise Exception('The GP emulator is already in non-Vecchia mode.')
        self.vecch = self.kernel.vecch = False
        self.kernel.compute_stats()

    def update_xy(self, X, Y, reset=False):
        """Replace the training data of the emulator (gp.py:144-181); reset=True also puts the hyper-parameters back
        to their initial values."""
        if Y.ndim == 1 or X.ndim == 1:
            raise Exception('The input and output data have to be numpy 2d-arrays.')
        self._set_data(X, Y)
        self.n_data = self.X.shape[0]
        self.m = min(self.m, self.n_data - 1)
        self.update_kernel(reset_lengthscale=reset)
        if self.vecch:
            self.kernel.ord_nn()
        else:
            self.kernel.compute_stats()

    def update_kernel(self, reset_lengthscale):
        """Hand the current data to the node (gp.py:183-209)."""
        k = self.kernel
        if self.indices is not None:
            k.rep, k.W_diag, k.sum_residual = self.indices, self.W_diag, self.sum_residual
        else:
            k.rep = k.W_diag = k.sum_residual = None
        bind_private(k, 'input', self.X[:, k.input_dim])
        if k.connect is not None:
            if len(np.intersect1d(k.connect, k.input_dim)) != 0:
                raise Exception('The local input and global input should not have any overlap. Change input_dim or '
                                'connect so they do not have any common indices.')
            bind_private(k, 'global_input', self.X[:, k.connect])
        k.output = self.Y.copy()
        k.m = self.m
        k._stats = None
        if reset_lengthscale:
            h = k.para_path[0, :]
            k.scale, k.length, k.nugget = h[[0]], h[1:-1], h[[-1]]
        if k.prior_name == 'ref':
            k.compute_cl()

    def metric(self, x_cand, method='MICE', nugget_s=1., m=50, score_only=False):
        """Sequential-design criterion at the rows of x_cand (gp.py:271-324): ALM = predictive variance; MICE = that
        variance over the variance of a GP on the candidate set alone (functions.mice_var); VIGF = 4 s2 b + 2 s2^2 with b the
        squared gap to the nearest training output."""
        if method == 'ALM' or method == 'MICE':
            _, s2 = self.predict(x=x_cand, m=m)
            if method == 'MICE':
                from .emulation import emulator
                e = emulator.__new__(emulator)
                e.engine = self.kernel.engine
                s2 = s2 / e._mice_var(x_cand, x_cand, self.kernel, nugget_s).reshape(-1, 1)
            score = s2
        elif method == 'VIGF':
            if self.indices is not None:
                raise Exception('VIGF criterion is currently not applicable to GP emulators whose training data contain replicates.')
            eng = self.kernel.engine
            index = eng.nn_query(eng.tensor(x_cand), eng.tensor(self.X), 1).cpu().numpy().flatten()
            mu, s2 = self.predict(x=x_cand, m=m)
            bias = (mu - self.Y[index, :]) ** 2
            score = 4 * s2 * bias + 2 * s2 ** 2
        else:
            raise Exception("method must be 'ALM', 'MICE' or 'VIGF'.")
        if score_only:
            return score
        idx = np.argmax(score, axis=0)
        return idx, score[idx, 0]

    def pmetric(self, x_cand, method='MICE', nugget_s=1., m=50, score_only=False, chunk_num=None, core_num=None):
        """gp.py:224-290 (`chunk_num` / `core_num` are accepted and unused: the candidates run in parallel on the device)."""
        return self.metric(x_cand, method=method, nugget_s=nugget_s, m=m, score_only=score_only)

    def train(self):
        """One L-BFGS-B fit of the hyper-parameters (gp.py:211-216)."""
        self.kernel.maximise()
        if not self.vecch:
            self.kernel.compute_stats()

    def export(self):
        """The trained node as a one-element structure for `container` (gp.py:218-222)."""
        k = copy.deepcopy(self.kernel)
        k.engine = self.kernel.engine
        return [k]

    def predict(self, x, method='mean_var', sample_size=50, m=50):
        """(mean, variance) as (M x 1) arrays, or samples (M x sample_size)  (gp.py:412-453)."""
        if x.ndim == 1:
            raise Exception('The testing input has to be a numpy 2d-array')
        k = self.kernel
        z = None if k.connect is None else x[:, k.connect]
        k.pred_m = m
        mu, s2 = k.gp_prediction(x=x[:, k.input_dim], z=z)
        if method == 'mean_var':
            return mu.reshape(-1, 1), s2.reshape(-1, 1)
        if method == 'sampling':
            return np.random.normal(mu, np.sqrt(s2), size=(sample_size, len(x))).T
        raise Exception("method must be 'mean_var' or 'sampling'.")

    def ppredict(self, x, method='mean_var', sample_size=50, m=50, chunk_num=None, core_num=None):
        """gp.py:373-410 (`chunk_num` / `core_num` are accepted and unused)."""
        return self.predict(x, method=method, sample_size=sample_size, m=m)

    def loo(self, method='mean_var', sample_size=50, m=30):
        """Leave-one-out predictions at the training inputs (gp.py:326-371).  Dense mode: the closed form on the
        stored statistics, sigma2_i = scale / (R^-1)_ii, mu_i = y_i - (R^-1 y)_i / (R^-1)_ii.  Vecchia mode: every point
        from its m nearest other points."""
        k = self.kernel
        if self.vecch:
            # vecchia.py:656-674: each point from its m nearest other points.  That is the Vecchia prediction at the
            # point with itself struck from its neighbour list; only the point's own nugget differs (it carries the
            # replicate weight here, not in a prediction), which shifts the variance by scale*nugget*(W_ii - 1).
            e = k.engine
            Xs = e.tensor(self.X / k.length)
            NN = e.nn_query(Xs, Xs, m + 1)[:, 1:].contiguous()
            wd = np.ones(len(self.Y)) if self.indices is None else self.W_diag
            Xd = e.tensor(self.X)
            mu, s2 = e.vecchia_gp(k.name, Xd, Xd, NN, e.tensor(self.Y.reshape(-1)), k.scale[0], k.length, k.nugget[0],
                                  e.tensor(wd))
            mu = mu.cpu().numpy().reshape(-1, 1)
            s2 = s2.cpu().numpy().reshape(-1, 1) + k.scale[0] * k.nugget[0] * (wd.reshape(-1, 1) - 1.0)
        else:
            st = k._stats
            n = st['n']
            import torch
            d = torch.diagonal(st['Rinv'])[:n]
            s2 = (1.0 / d).cpu().numpy().reshape(-1, 1)
            mu = self.Y - k.Rinv_y[:, None] * s2
            s2 = k.scale * s2
        if self.indices is not None:
            mu, s2 = mu[self.indices, :], s2[self.indices, :]
        if method == 'mean_var':
            return mu, s2
        if method == 'sampling':
            return np.random.normal(mu.flatten(), np.sqrt(s2.flatten()), size=(sample_size, len(mu))).T
        raise Exception("method must be 'mean_var' or 'sampling'.")

    def ploo(self, method='mean_var', sample_size=50, m=30, core_num=None):
        return self.loo(method=method, sample_size=sample_size, m=m)
