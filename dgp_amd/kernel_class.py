"""GP node of the DGP hierarchy -- host mirror of dgpsi's `kernel` plugin surface.

Same constructor, attributes and method names as the reference class
(kernel_class.py:9-764) so that imputers / trainers / emulators written against
dgpsi find what they expect; every numerical method is a thin call into the HIP
library through dgp_amd.ops.Engine.  Nothing here computes a kernel matrix, a
factorisation or a prediction on the CPU.

Differences that do not change results: `llik` obtains
tr(K^-1 dK_p) and y^T K^-1 dK_p K^-1 y from one inverse and fused in-flight
reductions instead of one n x n solve per parameter; `compute_stats` keeps
R^-1 / R^-1 y on the device and never materialises Psexp.
"""
import numpy as np
from numpy.linalg import LinAlgError
from .ops import raise_not_pd
from scipy.optimize import minimize, Bounds

from .ops import default_engine
from . import dist as ddist

_KINDS = ('sexp', 'matern2.5')


def combine(*layers):
    """Stack layers (lists of nodes) into a DGP structure (kernel_class.py:766-780)."""
    return [layer for layer in layers]


class TrackedInputs:
    """`input` / `global_input` of a node as the reference has them -- plain, writable numpy arrays (its sampler writes
    node.input[:, idx] in place, imputation.py:160-202) -- with one addition: the node remembers which array it has HANDED
    OUT.  The sampler keeps device copies of the node inputs and recognises an unchanged array by identity instead of
    comparing megabytes on every call (0.3 ms per 3-MB array, a dozen per iteration at n = 50 000); an array somebody else
    holds a reference to can change behind that test, so from the moment it is read -- or assigned -- through the public
    attribute it is compared by value.  The library's own code reads `_input` / `_global_input` and binds its private
    copies there (a new object: not handed out).  Rounds 2-3 froze the arrays instead (flags.writeable = False) and in-place
    writes raised ValueError."""
    _input = None
    _global_input = None
    _input_esc = 0            # id() of the array last handed out / assigned from outside
    _global_input_esc = 0

    @property
    def input(self):
        a = self._input
        if a is not None:
            self._input_esc = id(a)
        return a

    @input.setter
    def input(self, v):
        self._input = v
        self._input_esc = id(v)

    @property
    def global_input(self):
        a = self._global_input
        if a is not None:
            self._global_input_esc = id(a)
        return a

    @global_input.setter
    def global_input(self, v):
        self._global_input = v
        self._global_input_esc = id(v)

    def _private(self, name):
        """Is the array bound to `name` ('input' / 'global_input') one nobody outside the library has seen?"""
        a = getattr(self, '_' + name)
        return a is not None and getattr(self, '_' + name + '_esc') != id(a)

    def __setstate__(self, st):   # (objects pickled by rounds 1-3 kept the arrays under the public names)
        for k in ('input', 'global_input'):
            if k in st:
                st['_' + k] = st.pop(k)
        self.__dict__.update(st)


def bind_private(nd, name, arr):
    """Bind a private array (one the library has just made: nobody else holds it) as node.input / node.global_input."""
    if isinstance(nd, TrackedInputs):
        setattr(nd, '_' + name, arr)
    else:
        setattr(nd, name, arr)   # (a user plugin node: plain attributes)


def peek(nd, name):
    """node.input / node.global_input for the library's own reads: does not count as handing the array out."""
    return getattr(nd, '_' + name) if isinstance(nd, TrackedInputs) else getattr(nd, name)


class kernel(TrackedInputs):
    """A GP node.  Arguments as dgpsi.kernel (kernel_class.py:86); `engine` selects the
    device context (default: the process-wide engine of LOCAL_RANK)."""

    def __init__(self, length, scale=1., nugget=1e-6, name='sexp', prior_name='ga', prior_coef=None, bds=None,
                 nugget_est=False, scale_est=False, input_dim=None, connect=None, engine=None):
        if name not in _KINDS:
            raise Exception("name must be 'sexp' or 'matern2.5'.")
        self.type = 'gp'
        self.length = np.atleast_1d(np.asarray(length, dtype=float))
        self.scale = np.atleast_1d(np.asarray(scale, dtype=float))
        self.nugget = np.atleast_1d(np.asarray(nugget, dtype=float))
        self.name = name
        self.prior_name = prior_name
        # stored coefficients: ga -> (shape-1, rate); inv_ga -> (shape+1, scale); ref -> (a[, b])
        # (kernel_class.py:93-110; like the reference, a caller-supplied array is adjusted in place)
        if prior_name in ('ga', 'inv_ga'):
            self.prior_coef = np.array([1.6, 0.3]) if prior_coef is None else prior_coef
            self.prior_coef[0] += -1 if prior_name == 'ga' else 1
        elif prior_name == 'ref':
            self.prior_coef = np.array([0.2]) if prior_coef is None else prior_coef
            self.cl = None
        self.nugget_est, self.scale_est = nugget_est, scale_est
        self.input_dim, self.connect, self.bds = input_dim, connect, bds
        self.para_path = None
        self._global_input = self._input = self.output = None
        self.rep = self.rep_hetero = self.W_diag = self.sum_residual = None
        self.vecch = None
        self.D = None
        self.ord = self.rev_ord = self.NNarray = None
        self.m = self.pred_m = self.max_rep = None
        self.imp_NNarray = self.imp_pointer_row = self.imp_pointer_col = None
        self.nn_method, self.ord_fun = 'exact', None
        self.iter_count = 0
        self.target = 'dgp'
        self.R2 = None
        self.loo_state = False
        self._engine = engine
        self._stats = None       # device-side prediction statistics (compute_stats)
        self._staged = None

    # ------------------------------------------------------------------ plumbing
    @property
    def engine(self):
        if self._engine is None:
            self._engine = default_engine()
        return self._engine

    @engine.setter
    def engine(self, e):
        self._engine = e
        self._staged = None

    def __getstate__(self):
        st = dict(self.__dict__)
        st['_engine'] = None
        st['_staged'] = None
        st['_stats'] = None
        st.pop('_r2_cache', None)
        st.pop('_prestaged', None)
        st.pop('_vecch_cache', None)
        st.pop('_vecch_prestaged', None)
        st.pop('_dev_cache', None)
        return st

    def _dev_of(self, name, dtype=None):
        """Device copy of one of the node's index arrays (NNarray: 10 MB at n = 50 000; ord, rev_ord), uploaded once per
        array OBJECT: ord_nn() always binds new arrays, a strided checksum guards against writes in place."""
        import torch
        a = getattr(self, name)
        cache = self.__dict__.setdefault('_dev_cache', {})
        step = max(1, a.shape[0] // 61)
        key = (id(a), a.shape, int(np.asarray(a[::step], dtype=np.int64).sum()), id(self.engine))
        hit = cache.get(name)
        if hit is None or hit[0] != key or hit[2] is not a:
            hit = cache[name] = (key, self.engine.tensor(a, dtype=torch.int64 if dtype is None else dtype), a)
        return hit[1]

    def nn_dev(self):
        return self._dev_of('NNarray')

    def ord_dev(self):
        return self._dev_of('ord')

    def rev_ord_dev(self):
        return self._dev_of('rev_ord')

    def _X(self):
        return self._input if self._global_input is None else np.concatenate((self._input, self._global_input), 1)

    def _stage(self):
        """Upload the node's current numpy state (inputs, output, replicate weights)."""
        pre = self.__dict__.pop('_prestaged', None)   # (device views handed over by the imputer: dgp._m_step)
        if pre is not None:
            self._staged = pre
            return pre
        e = self.engine
        s = dict(Xl=e.tensor(self._input), Xg=None if self._global_input is None else e.tensor(self._global_input),
                 y=e.tensor(np.asarray(self.output, dtype=float).reshape(-1)),
                 W=None if self.rep is None else e.tensor(self.W_diag))
        self._staged = s
        return s

    def _raise_if_not_pd(self, info):
        bad = int(info)
        if bad != 0:
            raise_not_pd(bad)

    # ---------------------------------------------------------------- parameters
    def log_t(self):
        """log(lengthscales[, nugget]) -- the optimiser's variables (kernel_class.py:279-289)."""
        theta = np.concatenate((self.length, self.nugget)) if self.nugget_est else self.length
        return np.log(theta)

    def update(self, log_theta):
        theta = np.exp(log_theta)
        if self.nugget_est:
            self.length, self.nugget = theta[:-1], theta[[-1]]
        else:
            self.length = theta

    def compute_cl(self):
        """Scaling constant of the reference prior (kernel_class.py:207-225)."""
        n_out = len(self.output)
        if len(self.length) == 1:
            X = self._X()
            if self.vecch:
                rg = X.max(0) - X.min(0)
                self.cl = np.sqrt(rg @ rg) / n_out
            else:
                from scipy.spatial.distance import pdist
                self.cl = np.max(pdist(X, metric='euclidean')) / n_out
        else:
            X = self._X()
            self.cl = (X.max(0) - X.min(0)) / n_out ** (1 / len(self.length))

    def r2(self, overwritten=False):
        """R2 of the linear regression of `input` on `global_input` (kernel_class.py:227-243).  The design matrix is a
        constant of the node, so its rank test and an orthonormal basis of its column space are kept between calls
        (the reference redoes two SVD ranks and an SVD least-squares fit per M-step); the residual sums are
        |y - Q Q'y|^2, which is what lstsq returns for a full-column-rank design."""
        if self._global_input is None:
            return
        G = self._global_input
        sig = (id(G), G.shape, float(G[0, 0]), float(G[-1, -1]), float(G.sum()))
        hit = self.__dict__.get('_r2_cache')
        if hit is None or hit[0] != sig:
            Xd = np.concatenate((G, np.ones((len(G), 1))), axis=1)
            if np.linalg.matrix_rank(G) == np.linalg.matrix_rank(Xd):
                Xd = G
            Q = None
            if Xd.shape[0] > Xd.shape[1] and np.linalg.matrix_rank(Xd) == Xd.shape[1]:
                Q = np.linalg.qr(Xd)[0]
            hit = self._r2_cache = (sig, Xd, Q)
        _, Xd, Q = hit
        Xin = self._input
        if Xd.shape[0] == Xd.shape[1]:
            resid = np.zeros(Xin.shape[1])
        elif Q is not None:
            r = Xin - Q @ (Q.T @ Xin)
            resid = np.einsum('ij,ij->j', r, r)
        else:
            resid = np.linalg.lstsq(Xd, Xin, rcond=None)[1]
        rsq = 1 - resid / (len(Xin) * np.var(Xin, axis=0))
        self.R2 = np.atleast_2d(rsq) if overwritten else np.vstack((self.R2, rsq))

    def gfod(self, x):
        """Derivative of the gamma / inverse-gamma log prior with respect to log x (kernel_class.py:361-365)."""
        c = self.prior_coef
        return c[0] - c[1] * x if self.prior_name == 'ga' else -c[0] + c[1] / x

    def log_prior(self):
        """kernel_class.py:367-381."""
        c = self.prior_coef
        if self.prior_name == 'ref':
            t = np.sum(self.cl / self.length) + self.nugget
            return c[0] * np.log(t) - c[1] * t
        xs = np.concatenate((self.length, self.nugget)) if self.nugget_est else self.length
        if self.prior_name == 'ga':
            return np.sum(c[0] * np.log(xs) - c[1] * xs)
        return np.sum(-c[0] * np.log(xs) - c[1] / xs)

    def log_prior_fod(self):
        """d log prior / d log(parameters) (kernel_class.py:383-401)."""
        c = self.prior_coef
        if self.prior_name == 'ref':
            t = np.sum(self.cl / self.length) + self.nugget
            fod = (c[1] - c[0] / t) * self.cl / self.length
            if self.nugget_est:
                fod = np.concatenate((np.atleast_1d(fod), (c[0] / t - c[1]) * self.nugget))
            return np.atleast_1d(fod)
        xs = np.concatenate((self.length, self.nugget)) if self.nugget_est else self.length
        return c[0] - c[1] * xs if self.prior_name == 'ga' else -c[0] + c[1] / xs

    # ------------------------------------------------------------- kernel matrix
    def k_matrix(self, fod_eval=False):
        """Correlation matrix (and, if asked, its log-parameter derivatives) as numpy arrays
        (kernel_class.py:304-359).  The training loop never calls this; it exists for API parity."""
        e, s = self.engine, self._stage()
        K = e.kmatrix(self.name, s['Xl'], None, s['Xg'], self.length, self.nugget[0], W=s['W'])
        K = K.cpu().numpy()
        if not fod_eval:
            return K
        # derivative stack for callers that want it materialised: dK_p = c_p o K, elementwise from K itself
        X = self._X() / self.length
        n, D = X.shape
        P = 1 if len(self.length) == 1 else D
        fod = np.zeros((P, n, n))
        Koff = K.copy()
        np.fill_diagonal(Koff, 0.0)
        for d in range(D):
            r = np.abs(X[:, d][:, None] - X[:, d][None, :])
            if self.name == 'sexp':
                c = 2 * r * r
            else:
                e1, e2 = 1 + np.sqrt(5) * r, (5 / 3) * r * r
                c = e2 * e1 / (e1 + e2)
            fod[0 if P == 1 else d] += c * Koff
        if self.nugget_est:
            w = np.ones(n) if self.rep is None else self.W_diag
            fod = np.concatenate((fod, np.diag(self.nugget[0] * w)[None]), axis=0)
        return K, fod

    # ------------------------------------------------------------ log-likelihoods
    def log_likelihood_func(self):
        """ESS target (kernel_class.py:481-492)."""
        e, s = self.engine, self._stage()
        ll, info = e.loglik(self.name, s['Xl'], None, s['Xg'], self.length, self.nugget[0], self.scale[0], s['y'], W=s['W'])
        ll, info = ll.cpu().numpy(), info.cpu().numpy()
        self._raise_if_not_pd(info[0])
        out = ll[0]
        if self.prior_name == 'ref':
            self.compute_cl()
            out = out + self.log_prior()
        return np.atleast_2d(out) if np.ndim(out) == 0 else out

    def llik(self, x):
        """Negative log-likelihood and gradient wrt log-parameters (kernel_class.py:403-449)."""
        self.update(x)
        return self._llik_finish(self._llik_device())

    def _llik_device(self):
        """K -> Cholesky + inverse in one sweep (y as augmented row) -> in-flight derivative reductions for THIS node.
        Returns the host vector [logdet, y'K^-1y, tr_p.., quad_p.., info]."""
        e = self.engine
        s = self._staged if getattr(self, '_in_maximise', False) and self._staged is not None else self._stage()
        n = len(self.output)
        Np = e.padded_dim(n)
        A = e.workspace(('llikA', n), Np * Np * 8)
        Ainv = e.workspace(('llikAinv', n), Np * Np * 8)
        T = e.workspace(('llikT', n), Np * Np * 8)
        e.kmatrix(self.name, s['Xl'], None, s['Xg'], self.length, self.nugget[0], W=s['W'], out=A, full=False, Y=s['y'])
        logdet, info = e.potrf_inv(n, A, T, Ainv)   # factor, K^-1 (lower tiles) and -alpha in one sweep
        quad = e.aug_quad(n, A, 1, 1)
        red, P = e.grad_reduce(self.name, s['Xl'], None, s['Xg'], self.length, self.nugget[0], self.nugget_est, Ainv, W=s['W'])
        import torch
        return torch.cat((logdet, quad.reshape(-1), red, info.to(torch.float64))).cpu().numpy()

    def _llik_finish(self, host):
        """Closing host arithmetic of kernel.llik (kernel_class.py:421-448) on the reduced device results."""
        n = len(self.output)
        P = (len(host) - 3) // 2
        self._raise_if_not_pd(host[-1])
        if self.rep is None and self.prior_name in (None, 'ga', 'inv_ga') and P <= 2:
            # the same arithmetic in the same order on python floats (this runs once per node and L-BFGS-B round between
            # two device calls: a dozen numpy calls on one- and two-element arrays were most of the host's turn-around);
            # the logarithms stay numpy's (its vectorised log and libm's differ in the last bit now and then)
            logdet, YKinvY = float(host[0]), float(host[1])
            if self.scale_est:
                sc = YKinvY / n
                self.scale = np.array([sc])
                nll = 0.5 * (logdet + n * float(np.log(sc)))
            else:
                sc = float(self.scale[0])
                nll = 0.5 * (logdet + YKinvY / sc)
            g = [0.5 * float(host[2 + p]) - (0.5 * float(host[2 + P + p])) / sc for p in range(P)]
            if self.prior_name is not None:
                c0, c1 = float(self.prior_coef[0]), float(self.prior_coef[1])
                xs = [float(v) for v in self.length] + ([float(self.nugget[0])] if self.nugget_est else [])
                if len(xs) == P:
                    if self.prior_name == 'ga':
                        lp = [c0 * float(np.log(x)) - c1 * x for x in xs]
                        fod = [c0 - c1 * x for x in xs]
                    else:
                        lp = [-c0 * float(np.log(x)) - c1 / x for x in xs]
                        fod = [-c0 + c1 / x for x in xs]
                    nll = nll - (lp[0] if P == 1 else lp[0] + lp[1])
                    g = [g[p] - fod[p] for p in range(P)]
                    return np.array([nll]), np.array(g)
            else:
                return np.array([nll]), np.array(g)
        if self.rep is None and self.prior_name in (None, 'ga', 'inv_ga'):
            # several lengthscales, no replicates: the general path's numpy operations in the general path's order (same bits:
            # -(-0.5 tr) is 0.5 tr exactly, a division by a one-element array is a division by its element), without its
            # wrappers (atleast_1d, flatten, concatenate, np.sum's dispatch: 28 -> 13 numpy calls between two device calls)
            tr, ykky = host[2:2 + P], host[2 + P:2 + 2 * P]
            if self.scale_est:
                sc = host[1] / n
                self.scale = np.array([sc])
                nll = 0.5 * (host[0] + n * np.log(sc))
            else:
                sc = self.scale[0]
                nll = 0.5 * (host[0] + host[1] / sc)
            g = 0.5 * tr - (0.5 * ykky) / sc
            if self.prior_name is not None:
                c = self.prior_coef
                xs = np.concatenate((self.length, self.nugget)) if self.nugget_est else self.length
                if self.prior_name == 'ga':
                    nll = nll - np.add.reduce(c[0] * np.log(xs) - c[1] * xs)
                    g = g - (c[0] - c[1] * xs)
                else:
                    nll = nll - np.add.reduce(-c[0] * np.log(xs) - c[1] / xs)
                    g = g - (-c[0] + c[1] / xs)
            return np.array([nll]), g
        logdet, YKinvY, tr, ykky = host[0], host[1], host[2:2 + P], host[2 + P:2 + 2 * P]
        P1, P2 = -0.5 * tr, 0.5 * ykky
        rep = self.rep is not None
        if self.scale_est:
            if not rep:
                self.scale = np.atleast_1d(YKinvY / n)
                nll = 0.5 * (logdet + n * np.log(self.scale))
            else:
                self.scale = np.atleast_1d((YKinvY + self.sum_residual / self.nugget) / len(self.rep)).flatten()
                nll = 0.5 * (logdet + len(self.rep) * np.log(self.scale))
            g = -P1 - P2 / self.scale
            if rep and self.nugget_est:
                nll = nll + 0.5 * (len(self.rep) - n) * np.log(self.nugget)
                g[-1] += (0.5 * (-self.sum_residual / (self.scale * self.nugget) + (len(self.rep) - n)))[0]
        else:
            nll = 0.5 * (logdet + YKinvY / self.scale)
            g = -P1 - P2 / self.scale
            if rep and self.nugget_est:
                nll = nll + 0.5 * (self.sum_residual / (self.scale * self.nugget) + (len(self.rep) - n) * np.log(self.nugget))
                g[-1] += (0.5 * (-self.sum_residual / (self.scale * self.nugget) + (len(self.rep) - n)))[0]
        nll = np.atleast_1d(nll).flatten()
        if self.prior_name is not None:
            nll = nll - self.log_prior()
            g = g - self.log_prior_fod()
        return nll, g

    # ---------------------------------------------------------------- Vecchia twins
    def ord_nn(self, ord=None, NNarray=None, pointer=False, slot=None, rev_ord=None):
        """Ordering and ordered nearest neighbours (kernel_class.py:245-277); NN search on device.
        slot (a mailbox number, imputer.update_ord_nn): the search is launched and its result posted to the host; the call
        returns a function that waits for it and binds NNarray -- several nodes' searches then run back to back on the
        device while the host prepares the next one."""
        if ord is None:
            if self.ord_fun is None:
                self.ord = np.random.permutation(self._input.shape[0])
            else:
                self.ord = self.ord_fun(self._X() / self.length)
        else:
            self.ord = ord
        if rev_ord is not None:
            self.rev_ord = rev_ord
        else:   # (the inverse permutation: what np.argsort(self.ord) returns, without the sort)
            self.rev_ord = np.empty(len(self.ord), dtype=np.intp)
            self.rev_ord[self.ord] = np.arange(len(self.ord), dtype=np.intp)
        finish = None
        if NNarray is None:
            e = self.engine
            Xs = (self._X() / self.length)[self.ord]
            dev = e.nn_ordered(e.tensor(Xs), self.m)
            token = e.post(dev, 0 if slot is None else slot)   # (page-locked staging: a pageable copy of 10 MB takes 3.5 ms)

            def finish():
                self.NNarray = e.collect(token)
                self.__dict__.setdefault('_dev_cache', {}).pop('NNarray', None)
                step = max(1, self.NNarray.shape[0] // 61)   # (seed the device cache: no upload of what was just computed there)
                self._dev_cache['NNarray'] = ((id(self.NNarray), self.NNarray.shape, int(self.NNarray[::step].sum()), id(e)), dev, self.NNarray)
            if slot is None or pointer:
                finish()
                finish = None
        else:
            self.NNarray = NNarray
        if pointer:
            # conditioning sets of the LATENT values for a Hetero likelihood's exact-posterior step (kernel_class.py:268-274):
            # in the stacked vector [observations 0..n-1 ; latents n..2n-1] row i = own latent, own observation, the m-1
            # nearest other points (latent if earlier in the ordering, observation otherwise)
            e = self.engine
            Xs = e.tensor((self._X() / self.length)[self.ord])
            n = Xs.shape[0]
            NNs = e.nn_query(Xs, Xs, self.m)[:, 1:].cpu().numpy().copy()
            prev = NNs < np.arange(n)[:, None]
            NNs[prev] += n
            self.imp_NNarray = np.hstack((np.arange(n).reshape(-1, 1) + n, np.arange(n).reshape(-1, 1), NNs)).astype(np.int64)
        else:
            self.imp_NNarray = None
        return finish

    def _vecch_stage(self, trust_pre=False):
        """Ordered inputs / outputs, neighbour array and nugget weights on the device.  Kept between calls while nothing
        has changed (an L-BFGS-B run evaluates the same data dozens of times; the neighbour array alone is 10 MB at
        n = 50 000): the arrays' identities say whether they were replaced, sums whether they were written in place."""
        e = self.engine
        import torch
        if trust_pre and self.rep is None and self.__dict__.get('_vecch_prestaged') is not None:
            # (lock-step M-step: the imputer has just validated these views and the result is used for this run only --
            # no signature of the host arrays, 0.4 ms per node at n = 50 000)
            pre = self.__dict__.pop('_vecch_prestaged')
            od = self.ord_dev()
            ones = self.__dict__.setdefault('_dev_cache', {}).get('ones')
            if ones is None or ones.shape[0] != len(self.output) or ones.device != od.device:
                ones = self._dev_cache['ones'] = e.tensor(np.ones(len(self.output)))
            return dict(X=pre['X'][od].contiguous(), y=pre['y'][od].contiguous(), NN=self.nn_dev(), nd=ones)
        nd = np.ones(len(self.output)) if self.rep is None else self.W_diag
        sig = (id(self._input), id(self._global_input), id(self.ord), id(self.NNarray), id(nd) if self.rep is not None else None,
               float(np.sum(self._input)), float(np.sum(self.output)), float(np.sum(self.ord[:16])), len(self.output), id(e))
        hit = self.__dict__.get('_vecch_cache')
        pre = self.__dict__.pop('_vecch_prestaged', None)   # (device views of input and output handed over by the imputer)
        if hit is not None and hit[0] == sig:
            return dict(hit[1])
        if pre is not None and self.rep is None:
            od = self.ord_dev()
            ones = self.__dict__.setdefault('_dev_cache', {}).get('ones')
            if ones is None or ones.shape[0] != len(self.output) or ones.device != od.device:
                ones = self._dev_cache['ones'] = e.tensor(nd)
            st = dict(X=pre['X'][od].contiguous(), y=pre['y'][od].contiguous(), NN=self.nn_dev(), nd=ones)
        else:
            X = self._X()[self.ord]
            st = dict(X=e.tensor(X), y=e.tensor(np.asarray(self.output, float).reshape(-1)[self.ord]), NN=self.nn_dev(), nd=e.tensor(nd))
        self._vecch_cache = (sig, st)
        return dict(st)

    def log_likelihood_func_vecch(self):
        """kernel_class.py:494-509 -> vecchia_llik (vecchia.py:164-180)."""
        e, s = self.engine, self._vecch_stage()
        lo, hi = ddist.vecchia_rows(s['NN'].shape[0])   # (all rows unless dist.split_training(rows=True) on several ranks)
        out = e.vecchia_llik(self.name, s['X'], s['y'], s['NN'][lo:hi], self.length, self.nugget[0], s['nd'])
        out = ddist.allreduce_sum_vector(out).cpu().numpy() if ddist.rows_split() else out.cpu().numpy()
        ll = -0.5 * (out[1] + out[0] / self.scale[0])
        if self.prior_name == 'ref':
            self.compute_cl()
            ll = ll + self.log_prior()
        return np.atleast_1d(ll)

    def llik_vecch(self, x):
        """kernel_class.py:451-479 -> vecchia_nllik (vecchia.py:182-242); the closing scale_est /
        replicate algebra (vecchia.py:224-241) runs here on the reduced sums."""
        self.update(x)
        o, P = self._llik_vecch_device()
        o = ddist.allreduce_sum_vector(o).cpu().numpy() if ddist.rows_split() else o.cpu().numpy()
        return self._llik_vecch_finish(o, P)

    def _llik_vecch_device(self):
        """The device part of llik_vecch at the node's current hyper-parameters: (sums on the device, P), no synchronisation
        (the lock-step M-step queues several nodes' evaluations before it fetches them all)."""
        e = self.engine
        s = self.__dict__.get('_vecch_fixed') if getattr(self, '_in_maximise', False) else None   # (staged once per optimiser run)
        if s is None:
            s = self._vecch_stage()
        if self.rep is not None:
            s = dict(s)
            s['nd'] = e.tensor(self.W_diag[self.ord])
        lo, hi = ddist.vecchia_rows(s['NN'].shape[0])
        return e.vecchia_nllik(self.name, s['X'], s['y'], s['NN'][lo:hi], self.length, self.nugget[0], s['nd'], self.nugget_est)

    def _llik_vecch_finish(self, o, P):
        """The closing scale_est / replicate algebra of llik_vecch (vecchia.py:224-241) on the reduced sums (host)."""
        n = len(self.output)
        if self.rep is None:
            origin_n, rr = n, -1.0
        else:
            origin_n, rr = len(self.rep), float(self.sum_residual[0])
        from .vecchia import nllik_close
        nll, g, scale = nllik_close(o, P, n, origin_n, rr, self.scale[0], self.nugget[0], self.scale_est, self.nugget_est)
        self.scale = np.atleast_1d(scale)
        nll = np.atleast_1d(nll)
        if self.prior_name is not None:
            nll = nll - self.log_prior()
            g = g - self.log_prior_fod()
        return nll, g

    def callback(self, xk):
        self.iter_count += 1
        if self.iter_count & (self.iter_count - 1) == 0:
            self.ord_nn()

    # --------------------------------------------------------------------- M-step
    def _opt_setup(self):
        """Start point, bounds and options of the node's L-BFGS-B run (kernel_class.py:516-560)."""
        x0 = self.log_t()
        p = len(x0)
        nl = p - 1 if self.nugget_est else p
        lb, ub = np.full(p, -np.inf), np.full(p, np.inf)
        bounded = False
        if self.bds is not None:
            with np.errstate(divide='ignore'):
                lb[:nl], ub[:nl] = np.log(self.bds[0]), np.log(self.bds[1])
            bounded = True
        elif self.prior_name == 'ref':
            ub[:nl] = 13.
            bounded = True
        if self.nugget_est:
            lb[-1] = np.log(1e-8)
            bounded = True
        opts = {'maxiter': 100, 'maxfun': int(max(30, 20 + 5 * self.D))}
        if self.vecch and self.target == 'gp' and len(self.length) != 1:
            opts = {'maxfun': int(max(50, 20 + 5 * self.D))}
        return x0, (lb if bounded else None), (ub if bounded else None), opts

    def maximise(self, method='L-BFGS-B'):
        """One M-step of the node: scipy L-BFGS-B over log-parameters driving the device objective
        (kernel_class.py:516-579; same bounds, maxiter and maxfun)."""
        x0, lb, ub, opts = self._opt_setup()
        kw = dict(method=method, jac=True)
        if lb is not None:
            kw['bounds'] = Bounds(lb, ub)
        fun = self.llik
        if self.vecch:
            fun = self.llik_vecch
            if self.target == 'gp' and len(self.length) != 1:
                kw['callback'] = self.callback
        if not self.vecch:   # (Vecchia objectives stage their own ordered arrays: _vecch_stage)
            self._stage()
        self._in_maximise = True
        try:
            minimize(fun, x0, options=opts, **kw)
        finally:
            self._in_maximise = False
            self.iter_count = 0
        self.add_to_path()

    def add_to_path(self):
        self.para_path = np.vstack((self.para_path, np.concatenate((self.scale, self.length, self.nugget))))

    # ----------------------------------------------------------------- prediction
    def compute_stats(self):
        """R^-1 and R^-1 y for prediction (kernel_class.py:735-764), device resident.  When R is not numerically
        positive definite the pseudo-inverse takes over, as in the reference (:749-751)."""
        e, s = self.engine, self._stage()
        n = len(self.output)
        Np = e.padded_dim(n)
        A = e.workspace(('llikA', n), Np * Np * 8)
        Ainv = e.empty(Np, Np)
        e.kmatrix(self.name, s['Xl'], None, s['Xg'], self.length, self.nugget[0], W=s['W'], out=A, full=False, Y=s['y'])
        work = e.potrf_workspace(n, 1)
        _, info = e.potrf(n, A, work=work)
        e.potri(n, A, Ainv, 1, work)
        if int(info.cpu().numpy()[0]):
            K = e.kmatrix(self.name, s['Xl'], None, s['Xg'], self.length, self.nugget[0], W=s['W'], full=True)
            Ainv.zero_()
            Ainv[:n, :n] = e.pinvh(K)
            ry = e.gemv(Ainv[:n, :n], s['y'])
        else:
            ry = (-Ainv[n, :n]).contiguous()
        self._stats = dict(Rinv=Ainv, ld=Np, ry=ry, W=e.tensor(self._input),
                           Wg=None if self._global_input is None else e.tensor(self._global_input),
                           Wall=e.tensor(self._X()), n=n)

    @property
    def Rinv(self):
        return None if self._stats is None else self._stats['Rinv'][:self._stats['n'], :self._stats['n']].cpu().numpy()

    @property
    def Rinv_y(self):
        return None if self._stats is None else self._stats['ry'].cpu().numpy()

    @property
    def R2sexp(self):
        """exp(-sqdist/2) on the scaled local inputs (kernel_class.py:752-763); built on demand, never used internally."""
        if self.name != 'sexp' or self._input is None:
            return None
        e = self.engine
        ll = self.length if len(self.length) == 1 else self.length[:self._input.shape[1]]
        K = e.kmatrix('sexp', e.tensor(self._input), None, None, ll * np.sqrt(2.0), 0.0).cpu().numpy()
        np.fill_diagonal(K, 1.0)
        return K

    @property
    def Psexp(self):
        if self.name != 'sexp' or self._input is None:
            return None
        ll = self.length if len(self.length) == 1 else self.length[:self._input.shape[1]]
        Xl = self._input / ll
        return np.stack([Xl[:, d][:, None] + Xl[:, d][None, :] for d in range(Xl.shape[1])])

    def _pred_nn(self, x, w):
        """Conditioning sets of the test rows (vecchia.get_pred_nn, vecchia.py:20-40): the pred_m nearest training
        points, nearest first -- or, when pred_m covers all n of them, the reference's shortcut: row k = k, k+1, ...
        cyclically, no search (so that the leave-one-out walk of a dense emulator drops training point k for test
        row k, emulation.py:90-143)."""
        e = self.engine
        n = len(w)
        given = self.__dict__.pop('_nn_given', None)   # (handed over by the emulator: a sibling's search, see _layer_moments_vecchia)
        if given is not None and given.shape[0] == len(x) and given.shape[1] == min(self.pred_m, n) - (1 if self.loo_state else 0):
            return given
        if self.pred_m >= n:
            import torch
            NN = ((torch.arange(n, device=e.device)[None, :] + torch.arange(len(x), device=e.device)[:, None]) % n).contiguous()
        else:
            NN = e.nn_query(e.tensor(x / self.length), e.tensor(w / self.length), self.pred_m)
        if self.loo_state:
            NN = NN[:, 1:].contiguous()
        return NN

    def gp_prediction(self, x, z):
        """Mean/variance at deterministic inputs (kernel_class.py:587-625)."""
        e = self.engine
        xa = x if z is None else np.concatenate((x, z), 1)
        if self.vecch:
            w = self._X()
            nd = np.ones(len(self.output)) if self.rep is None else self.W_diag
            m, v = e.vecchia_gp(self.name, e.tensor(xa), e.tensor(w), self._pred_nn(xa, w),
                                e.tensor(np.asarray(self.output, float).reshape(-1)), self.scale[0], self.length,
                                self.nugget[0], e.tensor(nd))
        else:
            st = self._stats
            m, v = e.gp_predict(self.name, e.tensor(xa), st['Wall'], self.length, st['Rinv'], st['ld'], st['ry'],
                                self.scale[0], self.nugget[0])
        return m.cpu().numpy(), v.cpu().numpy()

    def linkgp_prediction(self, m, v, z):
        """Mean/variance at Gaussian-distributed inputs (kernel_class.py:627-670)."""
        e = self.engine
        zt = None if z is None else e.tensor(z)
        if self.vecch:
            x = m if z is None else np.concatenate((m, z), 1)
            w = self._X()
            nd = np.ones(len(self.output)) if self.rep is None else self.W_diag
            mo, vo = e.vecchia_linkgp(self.name, e.tensor(m), e.tensor(v), zt, e.tensor(self._input),
                                      None if self._global_input is None else e.tensor(self._global_input),
                                      self._pred_nn(x, w), e.tensor(np.asarray(self.output, float).reshape(-1)),
                                      self.scale[0], self.length, self.nugget[0], e.tensor(nd))
        else:
            st = self._link_stats()
            mo, vo = e.linkgp_predict(self.name, e.tensor(m), e.tensor(v), zt, st['W'], st['Wg'], self.length, st['Rinv'],
                                      st['ld'], st['ry'], self.scale[0], self.nugget[0])
        return mo.cpu().numpy(), vo.cpu().numpy()

    def _link_stats(self):
        """compute_stats' arrays as the linked predictor takes them: for a Matern-2.5 node a copy with the training points
        grouped by cells of the local inputs (Engine.linkgp_cells; built at the first linked prediction, dropped with the
        statistics), else the statistics themselves."""
        st = self._stats
        if 'link' not in st:
            cells = self.engine.linkgp_cells(self.name, self._input, st['Wg'], st['Rinv'], st['ry'])
            st['link'] = st if cells is None else dict(cells, ld=st['ld'])
        return st['link']

    def linkgp_prediction_full(self, m, v, m_z, v_z, z):
        """Linked prediction when part of the node's GLOBAL input is itself uncertain (outputs of feeding
        emulators): those columns join the Gaussian inputs, the rest stays deterministic (kernel_class.py:672-733).
        No R2sexp/Psexp bookkeeping is needed here -- the device kernel never uses them."""
        e = self.engine
        nz = m_z.shape[1]
        mm, vv = np.concatenate((m, m_z), axis=1), np.concatenate((v, v_z), axis=1)
        W = np.concatenate((self._input, self._global_input[:, :nz]), axis=1)
        Wg = self._global_input[:, nz:]
        zt = None if z is None else e.tensor(z)
        Wgt = None if z is None else e.tensor(Wg)
        if self.vecch:
            x = mm if z is None else np.concatenate((mm, z), 1)
            w = self._X()
            nd = np.ones(len(self.output)) if self.rep is None else self.W_diag
            mo, vo = e.vecchia_linkgp(self.name, e.tensor(mm), e.tensor(vv), zt, e.tensor(W), Wgt, self._pred_nn(x, w),
                                      e.tensor(np.asarray(self.output, float).reshape(-1)), self.scale[0], self.length,
                                      self.nugget[0], e.tensor(nd))
        else:
            st = self._stats
            hit = st.get('link_full')   # (Matern: the training points grouped by cells of the widened local input, built once)
            if hit is None or hit[0] != nz:
                cells = e.linkgp_cells(self.name, W, Wgt, st['Rinv'], st['ry'])
                hit = st['link_full'] = (nz, cells if cells is not None else dict(W=e.tensor(W), Wg=Wgt, Rinv=st['Rinv'], ry=st['ry']))
            c = hit[1]
            mo, vo = e.linkgp_predict(self.name, e.tensor(mm), e.tensor(vv), zt, c['W'], c['Wg'], self.length, c['Rinv'],
                                      st['ld'], c['ry'], self.scale[0], self.nugget[0])
        return mo.cpu().numpy(), vo.cpu().numpy()
