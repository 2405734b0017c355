"""Stochastic-EM trainer of a deep GP -- host driver mirroring dgpsi.dgp
(dgp.py:71-129 construction, :154-691 wiring, :1364-1412 train, :1517-1541 estimate).

The driver is plain Python like the reference's; what it drives is not: the I-step
is the device-resident ESS imputer (dgp_amd.imputation) and the M-step advances
the L-BFGS-B runs of all GP nodes in lock-step, their objective evaluations
(kernel.llik) batched on the device (dgp_amd.mstep).  Optimising the nodes side by
side instead of one after another does not change any result: given the imputed
latents every node's objective involves only its own hyper-parameters
(dgp.py:1391-1398).
"""
import copy
import os
import numpy as np
import torch
from tqdm import trange, tqdm

from .kernel_class import kernel as ker, combine, bind_private, peek, TrackedInputs
from . import dist as ddist
from .imputation import imputer, DrawStream
from .ops import Engine, default_engine, HandoffError
from . import utils


class dgp:
    """Args as dgpsi.dgp (dgp.py:71): X (n x d), Y (n x q), all_layer, check_rep, block, vecchia, m, ord_fun;
    plus `seed` (sampler streams) and `device`."""

    def __init__(self, X, Y, all_layer=None, check_rep=True, block=True, vecchia=False, m=25, ord_fun=None, seed=None,
                 device=None):
        if isinstance(Y, list):
            if len(Y) != 1:
                raise Exception('Y has to be a numpy 2d-array rather than a list. The list version of Y (for linked '
                                'emulation) has been reduced. Please use the dedicated lgp class for linked emulation.')
            Y = Y[0]
        if Y.ndim == 1 or X.ndim == 1:
            raise Exception('The input and output data have to be numpy 2d-arrays.')
        self.Y = Y
        self.check_rep = check_rep
        self.indices = None
        self.counts = None
        self.X = X
        if check_rep:
            X0, inv, counts = np.unique(X, return_inverse=True, return_counts=True, axis=0)
            if len(X0) != len(X):
                self.X, self.indices, self.counts = X0, np.asarray(inv).reshape(-1), counts
        self.vecch = vecchia
        self.n_data = self.X.shape[0]
        self.nn_method = 'exact'
        self.m = min(m, self.n_data - 1)
        self.ord_fun = ord_fun
        self.engine = default_engine(device)
        if all_layer is None:
            D, q = self.X.shape[1], self.Y.shape[1]
            all_layer = combine([ker(length=np.array([1.])) for _ in range(D)],
                                [ker(length=np.array([1.]), scale_est=True, connect=np.arange(D)) for _ in range(q)])
        self.all_layer = all_layer
        self.n_layer = len(all_layer)
        for l, layer in enumerate(all_layer):
            for nd in layer:
                if nd.type == 'likelihood' and (nd.name not in ('Hetero', 'Poisson', 'NegBin', 'ZIP', 'ZINB', 'Categorical') or l != self.n_layer - 1):
                    raise NotImplementedError('likelihood nodes belong in the final layer (Hetero, Poisson, NegBin, ZIP, ZINB, Categorical)')
        self._encode_labels(fit=True)
        self.initialize()
        self.block = block
        self.draws = DrawStream(seed)
        self.imp = imputer(self.all_layer, self.block, draws=self.draws, engine=self.engine)
        with self._init_scale():
            self.imp.sample(burnin=10)
        self.compute_r2()
        self.N = 0
        self.burnin = None

    def _encode_labels(self, fit=False):
        """Categorical likelihood: class labels -> 0..K-1 (dgp.py:112-122,843-844); K and the link default from the data."""
        lik = self.all_layer[-1][0]
        if getattr(lik, 'name', None) != 'Categorical':
            return
        if fit:
            from sklearn.preprocessing import LabelEncoder
            lik.class_encoder = LabelEncoder()
            self.Y = lik.class_encoder.fit_transform(self.Y.flatten()).reshape(-1, 1)
            if lik.num_classes is None:
                lik.num_classes = len(lik.class_encoder.classes_)
            if lik.link is None:
                lik.link = 'logit' if lik.num_classes == 2 else 'softmax'
        else:
            self.Y = lik.class_encoder.transform(self.Y.flatten()).reshape(-1, 1)

    def _init_scale(self):
        """Context of the first sweeps under a Categorical likelihood (dgp.py:1574-1589): the feeding nodes whose variance
        is estimated sample with variance 40 (the warm-start latents are +-2 sqrt(40)), restored afterwards."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old = []
            cat = getattr(self.all_layer[-1][0], 'name', None) == 'Categorical'
            if cat:
                for nd in self.all_layer[-2]:
                    old.append(nd.scale)
                    if nd.scale_est:
                        nd.scale = np.array([40.0])
            try:
                yield
            finally:
                if cat:
                    for o, nd in zip(old, self.all_layer[-2]):
                        nd.scale = o
        return ctx()

    def _categorical_warm_start(self, l):
        """Latents feeding a Categorical likelihood (dgp.py:279-326), or None: +-2 sqrt(40) on the observed class without
        replicates; smoothed empirical logits per site with them."""
        if l != self.n_layer - 2 or len(self.all_layer[l + 1]) != 1 or getattr(self.all_layer[l + 1][0], 'name', None) != 'Categorical':
            return None
        lik, M = self.all_layer[l + 1][0], len(self.all_layer[l])
        K = lik.num_classes
        if (K == 2 and M != 1) or (K != 2 and M != K):
            raise Exception('You need %s to feed the Categorical likelihood node.' % ('one GP node' if K == 2 else '%d GP nodes' % K))
        c = 2 * np.sqrt(40)
        y = self.Y.ravel().astype(int)
        if self.indices is None:
            if K == 2:
                return np.where(self.Y == 1, c, -c).astype(float)
            Out = -c * np.ones((self.n_data, K))
            Out[np.arange(self.n_data), y] = c
            return Out
        G = self.indices.max() + 1
        eps = np.finfo(float).eps
        if K == 2:
            n_g = np.bincount(self.indices, minlength=G)
            p = (np.bincount(self.indices, weights=y, minlength=G) + 0.5) / (n_g + 1.0)
            return np.log(np.clip(p, eps, 1 - eps) / np.clip(1 - p, eps, 1)).reshape(-1, 1)
        counts = np.zeros((G, K))
        np.add.at(counts, (self.indices, y), 1.0)
        logp = np.log(((counts + 0.5) / (counts.sum(axis=1, keepdims=True) + K * 0.5)).clip(eps, 1.0))
        logp -= logp.mean(axis=1, keepdims=True)
        return logp / 0.8

    # ------------------------------------------------------------------ wiring
    def _warm_start(self, In, num_kernel):
        """Latent initial values of a hidden layer (dgp.py:565-576)."""
        d = In.shape[1]
        if d == num_kernel:
            return copy.copy(In)
        if d > num_kernel:
            if self.vecch or self.n_data >= 500:
                return utils.NystromKPCA(n_components=num_kernel).fit_transform(In)
            from sklearn.decomposition import KernelPCA
            return KernelPCA(n_components=num_kernel, kernel='sigmoid').fit_transform(In)
        return np.concatenate((In, In[:, np.random.choice(d, num_kernel - d)]), 1)

    # ---- warm starts of the latents under a count likelihood (dgp.py:327-566): moment estimates per replicated site.
    #      Pinned on the reference's own values (tests/golden/g17_count_likelihoods.npz).
    _TINY, _RATE_FLOOR, _ZI_LO, _ZI_HI = 1e-12, 1e-6, 1e-4, 0.99

    def _site_sums(self, y, G):
        """(count, sum y, sum y^2, number of zeros) per input site."""
        idx = np.asarray(self.indices)
        tally = lambda w=None: np.bincount(idx, weights=w, minlength=G)
        return tally().astype(float), tally(y), tally(y * y), tally((y == 0).astype(float))

    @staticmethod
    def _dispersion(mean, var, fallback=None, eps=1e-8):
        """Method of moments for Var = mu + sigma mu^2; where it fails (non-positive, not finite) the fallback."""
        sig = (var - mean) / (mean ** 2 + eps)
        if fallback is None:
            return sig
        sig = np.where(np.isfinite(sig) & (sig > 0.0), sig, fallback)
        return np.clip(sig, 1e-3, 10.0)

    @staticmethod
    def _site_variance(cnt, s1, s2, default):
        """Unbiased within-site variance where a site has replicates, else `default`."""
        out = np.array(default, dtype=float, copy=True)
        rep = cnt > 1
        out[rep] = (s2[rep] - s1[rep] ** 2 / cnt[rep]) / (cnt[rep] - 1.0)
        return out

    @classmethod
    def _zero_excess_logit(cls, p_zero, q_zero):
        """logit of the share of zeros beyond what the count model explains: (p0 - q0) / (1 - q0), kept inside (0, 1)."""
        pi = np.clip(np.where(p_zero <= q_zero, 0.0, (p_zero - q_zero) / np.maximum(1.0 - q_zero, 1e-8)), 0.0, cls._ZI_HI)
        pi = np.clip(pi, cls._ZI_LO, 1.0 - cls._ZI_LO)
        return pi, np.log(pi / (1.0 - pi))

    @classmethod
    def _zero_excess_global(cls, y):
        """The same from the whole sample (no replicates): a scalar logit."""
        p0 = ((y == 0).sum() + 0.5) / (len(y) + 1.0)
        mu = y.mean()
        if mu <= 0:
            pi = p0
        else:
            q0 = np.exp(-max(mu, cls._RATE_FLOOR))
            pi = 0.0 if q0 >= 1.0 - 1e-8 else np.clip((p0 - q0) / (1.0 - q0), 0.0, cls._ZI_HI)
        pi = np.clip(pi, cls._ZI_LO, 1.0 - cls._ZI_LO)
        return np.log(pi / (1.0 - pi))

    def _count_warm_start(self, l):
        """Initial latents of the layer feeding a lone Poisson / NegBin / ZIP / ZINB likelihood (dgp.py:327-566), or None:
        log of the (site-pooled) counts + 1/2; a log-dispersion from the method of moments per replicated site; the logit
        of the zero excess over a Poisson with the observed mean."""
        if l != self.n_layer - 2 or len(self.all_layer[l + 1]) != 1:
            return None
        name = getattr(self.all_layer[l + 1][0], 'name', None)
        if name not in ('Poisson', 'NegBin', 'ZIP', 'ZINB'):
            return None
        y = self.Y.flatten()
        G, M = self.X.shape[0], len(self.all_layer[l])
        pooled = self.indices is not None
        Out = np.empty((G, M))
        if name in ('Poisson', 'NegBin'):
            if pooled:
                cnt, s1, s2, _ = self._site_sums(y, G)
                mu = (s1 + .5) / cnt
            else:
                mu = y + .5
            if name == 'Poisson':
                return np.log(mu + self._TINY).reshape(G, 1) if pooled else np.log(self.Y + .5 + self._TINY)
            Out[:, 0] = np.log(mu + self._TINY)
            glob = max(self._dispersion(y.mean(), y.var(ddof=1)), 1e-3)
            if pooled:
                Out[:, 1] = np.log(self._dispersion(mu, self._site_variance(cnt, s1, s2, mu), glob))
            elif M > 1:   # (the reference leaves this latent uninitialised without replicates, dgp.py:528-531: here the global estimate)
                Out[:, 1:] = np.log(glob)
            return Out
        # zero-inflated: ZIP = (log rate, logit pi), ZINB = (log mean, log dispersion, logit pi)
        zi = 1 if name == 'ZIP' else 2
        if name == 'ZINB':
            ym = y.mean()
            glob = min(max(self._dispersion(ym, y.var(ddof=1)) if y.size > 1 else 1.0, 1e-3), 10.0)
        if not pooled:
            Out[:, 0] = np.log(np.maximum(y + 0.5, self._RATE_FLOOR) + self._TINY)
            if name == 'ZINB':
                Out[:, 1] = np.log(glob)
            Out[:, zi] = self._zero_excess_global(y)
            return Out
        cnt, s1, s2, zeros = self._site_sums(y, G)
        raw = s1 / np.maximum(cnt, 1.0)
        p_zero = (zeros + 0.1) / (cnt + 0.2)
        rate0 = raw.copy()
        rate0[raw == 0.0] = y[y > 0].mean() if np.any(y > 0) else 1.0
        rate0 = np.maximum(rate0, self._RATE_FLOOR)
        pi, logit = self._zero_excess_logit(p_zero, np.exp(-rate0))
        Out[:, zi] = logit
        if name == 'ZIP':
            pi_raw = np.clip(np.where(p_zero <= np.exp(-rate0), 0.0, (p_zero - np.exp(-rate0)) / np.maximum(1.0 - np.exp(-rate0), 1e-8)),
                             0.0, self._ZI_HI)   # (before the lower clip: the rate is deflated by the unclipped share)
            rate = np.maximum(np.where(raw == 0.0, rate0, raw / np.maximum(1.0 - pi_raw, 1e-3)), self._RATE_FLOOR)
            Out[:, 0] = np.log(rate + self._TINY)
            return Out
        mu = (s1 + 0.5) / np.maximum(cnt, 1.0)
        Out[:, 0] = np.log(mu + self._TINY)
        Out[:, 1] = np.log(self._dispersion(mu, self._site_variance(cnt, s1, s2, mu), glob))
        return Out

    def _layer_warm_start(self, l, In, num_kernel):
        if self._is_hetero_pair(l):
            return self._hetero_warm_start()
        cnt = self._count_warm_start(l)
        if cnt is None:
            cnt = self._categorical_warm_start(l)
        return cnt if cnt is not None else self._warm_start(In, num_kernel)

    def _is_hetero_pair(self, l):
        """Layer l feeds a lone Hetero likelihood with exactly two GP nodes (dgp.py:163)."""
        return (l == self.n_layer - 2 and len(self.all_layer[l]) == 2 and len(self.all_layer[l + 1]) == 1
                and getattr(self.all_layer[l + 1][0], 'name', None) == 'Hetero')

    def _hetero_warm_start(self):
        """Initial (mean, log-variance) latents under a Hetero likelihood (dgp.py:163-246).  Without replicates: a
        reference-prior GP is fitted to y, its leave-one-out residuals give log-variance targets, a second GP fitted to
        those is sampled (clipped to +-2.576 sd).  With replicates: site means and bias-corrected log sample
        variances; sites observed once take the median variance first and are then redrawn from a GP fitted to the
        log-variance targets."""
        from .gp import gp as single_gp
        from scipy.special import psi
        G, D = self.X.shape
        y = self.Y.flatten()
        Out = np.empty((G, 2))
        if self.indices is None:
            Out[:, 0] = y
            fit = single_gp(self.X, y.reshape(-1, 1),
                            ker(length=np.ones(D), name=self.all_layer[-2][0].name, scale_est=True, nugget_est=True,
                                prior_name='ref', nugget=1e-2), vecchia=self.vecch, m=self.m, ord_fun=self.ord_fun)
            fit.train()
            m_mu = fit.loo()[0].flatten()
            z = np.log(np.maximum((y - m_mu) ** 2, 1e-12) + 1e-12)
            fit = single_gp(self.X, z.reshape(-1, 1),
                            ker(length=np.ones(D), name=self.all_layer[-2][1].name, scale_est=True, nugget_est=True,
                                prior_name='ref', nugget=1e-2), vecchia=self.vecch, m=self.m, ord_fun=self.ord_fun)
            fit.train()
            m_lv, v_lv = fit.loo()
            m_lv = m_lv.flatten()
            sd = np.sqrt(np.maximum((v_lv - fit.kernel.nugget * fit.kernel.scale).flatten(), 1e-12))
            Out[:, 1] = np.clip(np.random.normal(loc=m_lv, scale=sd), m_lv - 2.576 * sd, m_lv + 2.576 * sd)
        else:
            counts = np.bincount(self.indices, minlength=G).astype(float)
            s1 = np.bincount(self.indices, weights=y, minlength=G)
            s2sum = np.bincount(self.indices, weights=y * y, minlength=G)
            Out[:, 0] = s1 / counts
            valid = counts > 1.0
            s2 = np.full(G, np.nan)
            s2[valid] = np.maximum((s2sum - s1 ** 2 / np.maximum(counts, 1.0))[valid] / (counts[valid] - 1.0), 0.0)
            s2 = np.where(valid, s2, np.nanmedian(s2[valid]))
            nu = (counts - 1.0) / 2.0
            with np.errstate(divide='ignore', invalid='ignore'):
                bias = np.where(valid, psi(nu) - np.log(np.maximum(nu, 1e-12)), 0.0)
            z_init = np.log(s2 + 1e-12) - bias
            if np.any(~valid):
                # sites observed once: a (slightly relaxed) GP on the log-variance targets fills them in from their
                # leave-one-out predictive distributions, clipped to +-2 sd (dgp.py:246-258)
                fit = single_gp(self.X, z_init.reshape(-1, 1),
                                ker(length=np.ones(D) * 2., name=self.all_layer[-2][1].name, scale_est=True, nugget_est=True,
                                    prior_name='ref', nugget=1e-1), vecchia=self.vecch, m=self.m, ord_fun=self.ord_fun)
                fit.train()
                m_lv, v_lv = fit.loo()
                sing = ~valid
                m_s = m_lv[sing].flatten()
                sd_s = np.sqrt(np.maximum((v_lv[sing] - fit.kernel.nugget * fit.kernel.scale).flatten(), 1e-12))
                z_init = z_init.copy()
                z_init[sing] = np.clip(np.random.normal(loc=m_s, scale=sd_s), m_s - 2 * sd_s, m_s + 2 * sd_s)
            Out[:, 1] = z_init
        lik_dim = self.all_layer[-1][0].input_dim
        if lik_dim is not None:
            Out = Out[:, lik_dim]
        return Out

    def initialize(self):
        """Assign input / global_input / output / D / para_path to every node (dgp.py:154-691; GP nodes and a Hetero
        likelihood in the final layer)."""
        global_in = In = self.X
        for l, layer in enumerate(self.all_layer):
            last = l == self.n_layer - 1
            Out = None if last else self._layer_warm_start(l, In, len(layer))
            for k, nd in enumerate(layer):
                if last and self.indices is not None:
                    nd.rep = self.indices
                if nd.input_dim is None:
                    nd.input_dim = np.arange(In.shape[1])
                if nd.type == 'likelihood':
                    need = {'Poisson': 1, 'ZINB': 3}.get(nd.name, 2)
                    if nd.name == 'Categorical':
                        need = len(nd.input_dim)
                    if len(nd.input_dim) != need:
                        raise Exception(('You need one and only one GP node', 'You need two and only two GP nodes',
                                         'You need three and only three GP nodes')[need - 1]
                                        + ' to feed the ' + nd.name + ' likelihood node.')
                    bind_private(nd, 'input', In[nd.rep, :][:, nd.input_dim] if nd.rep is not None else In[:, nd.input_dim])
                    nd.output = self.Y[:, [k]]
                    continue
                bind_private(nd, 'input', In[:, nd.input_dim])
                if nd.type == 'gp':
                    if nd.connect is not None:
                        if l == 0 and len(np.intersect1d(nd.connect, nd.input_dim)) != 0:
                            raise Exception('The local input and global input should not have any overlap. Change '
                                            'input_dim or connect so they do not have any common indices.')
                        bind_private(nd, 'global_input', global_in[:, nd.connect])
                    nd.vecch, nd.m, nd.nn_method = self.vecch, self.m, self.nn_method
                    if self.ord_fun is not None:
                        nd.ord_fun = self.ord_fun
                    nd.D = peek(nd, 'input').shape[1] + (0 if nd.connect is None else len(nd.connect))
                    nd.engine = self.engine
                if last:
                    if nd.type == 'gp' and nd.rep is not None:
                        G = nd.rep.max() + 1
                        cnt = np.bincount(nd.rep, minlength=G)
                        nd.W_diag = 1.0 / cnt
                        nd.output = (np.bincount(nd.rep, weights=self.Y[:, k], minlength=G) * nd.W_diag).reshape(-1, 1)
                        res = self.Y[:, [k]] - nd.output[nd.rep, :]
                        nd.sum_residual = (res.T @ res).flatten()
                    else:
                        nd.output = self.Y[:, [k]].copy()
                else:
                    nd.output = Out[:, [k]].copy()
                if nd.type == 'gp':
                    if nd.prior_name == 'ref':
                        p = nd.D
                        nd.prior_coef = np.concatenate((nd.prior_coef, 1 / len(nd.output) ** (1 / p) * (nd.prior_coef + p)))
                        nd.compute_cl()
                    nd.para_path = np.atleast_2d(np.concatenate((nd.scale, nd.length, nd.nugget)))
            if self.vecch:
                self._layer_ord_nn(layer)
            if not last:
                In = copy.copy(Out)

    def _layer_ord_nn(self, layer):
        imputer([layer], engine=self.engine).update_ord_nn()

    def to_vecchia(self, m=25, ord_fun=None):
        if self.vecch:
            raise Exception('The DGP structure is already in Vecchia mode.')
        self.vecch, self.m, self.ord_fun = True, min(m, self.n_data - 1), ord_fun
        for layer in self.all_layer:
            for nd in layer:
                if nd.type == 'gp':
                    nd.vecch, nd.m, nd.ord_fun = True, self.m, ord_fun
            self._layer_ord_nn(layer)

    def remove_vecchia(self):
        if not self.vecch:
            raise Exception('The DGP structure is already in non-Vecchia mode.')
        self.vecch = False
        for layer in self.all_layer:
            for nd in layer:
                if nd.type == 'gp':
                    nd.vecch = False

    def compute_r2(self):
        for layer in self.all_layer[1:]:
            for nd in layer:
                if nd.type == 'gp':
                    nd.r2(overwritten=True)

    # ------------------------------------------------------------------ training
    def _m_step(self, early=False):
        """One L-BFGS-B fit per GP node (dgp.py:1391-1398).  early (see _mstep_can_start_early): the imputer's detach and the
        R2 diagnostics run after the lock-step driver has queued its first round of evaluations.  The dense nodes' optimisers advance in lock-step from this
        thread, their objective evaluations batched on the device (dgp_amd.mstep); Vecchia nodes (and every node if
        scipy's L-BFGS-B core is not the expected one) run kernel.maximise() one after another like the reference."""
        from . import mstep
        eng = self.engine
        nodes = [(l, nd) for l, layer in enumerate(self.all_layer) for nd in layer if nd.type == 'gp']
        every = nodes
        split = ddist.nodes_split()
        if split:   # this rank fits nodes rank, rank + world, ...; the fits are exchanged below
            nodes = [nodes[i] for i in range(ddist.rank(), len(nodes), ddist.world())]
        failure = None
        with eng.stream():
            def diagnostics():
                for l, nd in every:   # (on EVERY rank, also for the nodes another rank fits: identical R2 histories)
                    nd.engine = eng
                    if nd.prior_name == 'ref':
                        nd.compute_cl()
                    if l != 0:
                        nd.r2()
            if early:
                for _, nd in every:
                    nd.engine = eng

                def hook():
                    self.imp.finish_detach()
                    diagnostics()
            else:
                hook = None
                diagnostics()
            try:
                self._fit_nodes(nodes, mstep, eng, hook)
            except KeyboardInterrupt:
                raise
            except BaseException as exc:   # noqa: BLE001  (LinAlgError: restart together; anything else -- DgpAmdError, an
                # optimiser's ValueError, MemoryError -- give up together: a rank that skipped the collective would leave
                # the others blocked in the all-gather)
                if not split:
                    raise
                failure = exc   # rank-local: the other ranks are on their way into the collective -- join it, then fail together
            if split:
                self._exchange_fits(every, failure)

    def _fit_nodes(self, nodes, mstep, eng, hook=None):
        dense = [nd for _, nd in nodes if not nd.vecch] if mstep._HAVE_CORE else []
        if hook is not None:
            pre = self.imp.stage_for_mstep(trusted=True)
            if any(id(nd) not in pre for _, nd in nodes):   # (cannot happen for the shapes _mstep_can_start_early admits)
                hook()
                hook = None
                pre = self.imp.stage_for_mstep()
        else:
            pre = self.imp.stage_for_mstep() if getattr(self, 'imp', None) is not None and mstep._HAVE_CORE else {}
        for _, nd in nodes:
            if id(nd) in pre:
                if nd.vecch:
                    nd._vecch_prestaged = pre[id(nd)]
                elif any(nd is d for d in dense):
                    nd._prestaged = pre[id(nd)]
        if hook is not None:   # (run once, wherever the first round of evaluations is launched -- or at the end if there was none)
            inner, called = hook, [False]

            def hook():
                if not called[0]:
                    called[0] = True
                    inner()
        if dense:
            self.last_mstep = mstep.maximise_lockstep(eng, dense, self, after_first_launch=hook)
            if hook is not None:
                hook()
        else:
            self.last_mstep = None
        # Vecchia nodes whose optimiser needs no callback: lock-step as well (one synchronisation per round, not one per
        # node and evaluation)
        vlock = [nd for _, nd in nodes if nd.vecch and mstep._HAVE_CORE and not (nd.target == 'gp' and len(nd.length) != 1)]
        if len(vlock) > 1:
            mstep.maximise_lockstep_vecch(eng, vlock, after_first_launch=hook)
        else:
            vlock = []
            if hook is not None:
                hook()
        for _, nd in nodes:
            if not any(nd is d for d in dense) and not any(nd is d for d in vlock):
                nd.maximise()

    def _exchange_fits(self, every, failure):
        """Node split: ONE equal-size all-gather of doubles per M-step.  Rank r sends, for each node it owns, the row
        [status, scale, nugget, lengthscales...] (status 1: this rank's fits raised); afterwards every rank holds every
        node's estimate -- or every rank raises the same LinAlgError, so that train()'s restart (dgp.py:1402-1412) happens
        on all of them together and the models stay identical (a non-numerical failure: RuntimeError everywhere)."""
        w, r = ddist.world(), ddist.rank()
        width = 3 + max(len(nd.length) for _, nd in every)
        per_rank = -(-len(every) // w)
        mine = np.zeros((per_rank, width))
        for slot, i in enumerate(range(r, len(every), w)):
            nd = every[i][1]
            mine[slot, 1], mine[slot, 2] = nd.scale[0], nd.nugget[0]
            mine[slot, 3:3 + len(nd.length)] = nd.length
        if failure is not None:   # 1: numerical (every rank restarts), 2: anything else (every rank gives up)
            mine[:, 0] = 1.0 if isinstance(failure, (np.linalg.LinAlgError, SystemError)) else 2.0
        got = ddist.allgather_vector(mine, device=getattr(self.engine, 'device', None)).reshape(w, per_rank, width)
        bad = [q for q in range(w) if got[q, :, 0].any()]
        if bad:
            # (the fits this rank did make stay applied; every rank raises, and train()'s restart -- reinit_all_layer with
            #  reset_lengthscale -- puts scale, nugget and lengthscales of ALL nodes back to para_path[row] on every rank)
            msg = 'M-step failed on rank(s) %s of the node split%s' % (bad, '' if failure is None else ': %s' % failure)
            if got[:, :, 0].max() >= 2.0:
                raise RuntimeError(msg) from failure
            raise np.linalg.LinAlgError(msg) from failure
        for i, (_, nd) in enumerate(every):
            q, slot = i % w, i // w
            if q == r:
                continue
            row = got[q, slot]
            nd.scale, nd.nugget = np.atleast_1d(row[1]).copy(), np.atleast_1d(row[2]).copy()
            nd.length = row[3:3 + len(nd.length)].copy()
            nd.add_to_path()

    def train(self, N=500, ess_burn=10, disable=False):
        """N iterations of stochastic EM (dgp.py:1364-1412) with the same restart policy on LinAlgError."""
        N0, restarts, max_restarts = self.N, 0, 3
        self.__dict__.pop('_imp_trusted', None)   # (between two train() calls the caller may have edited the nodes' arrays)
        while True:
            pgb = None
            try:
                pgb = trange(1, N + 1, disable=disable)
                for i in pgb:
                    # (what an interrupted iteration may already have appended: every node's path length at the start)
                    marks = [(nd, None if nd.para_path is None else len(nd.para_path)) for layer in self.all_layer for nd in layer if nd.type == 'gp']
                    for attempt in (0, 1):
                        try:
                            self._si_iteration(i, ess_burn)
                            break
                        except HandoffError as exc:
                            # the one-launch factorisation lost a hand-off (its workgroups were not co-resident): once, the
                            # iteration is run again through the per-block-step kernel, which has no in-kernel waits; the
                            # elliptical-slice transition is valid from wherever the interrupted I-step left the latents
                            if attempt or ddist.is_active():   # (ranks must not diverge: with a training split every rank raises)
                                raise
                            self._handoff_fallback(exc, marks)
                    pgb.set_description('Iteration %i: Layer %i' % (i, self.n_layer))
                self.N += N
                return
            except (np.linalg.LinAlgError, SystemError):
                restarts += 1
                if pgb is not None:
                    pgb.close()
                if restarts > max_restarts:
                    raise RuntimeError(f"Training failed after {max_restarts} restarts.")
                if not disable:
                    tqdm.write(f"Restart {restarts}/{max_restarts}:")
                self.N = N0
                self.reinit_all_layer(reset_lengthscale=True, row=self.N)

    def _si_iteration(self, i, ess_burn):
        """One iteration of stochastic EM: I-step, neighbour refresh, M-step (dgp.py:1377-1398)."""
        it = self.N + i
        refresh = self.vecch and (it & (it - 1)) == 0 and it > 1   # NN refresh at iterations 2,4,8,.. (dgp.py:1388)
        early = self._mstep_can_start_early()
        # the previous iteration of THIS train() call ended normally: the nodes' arrays are what the imputer itself wrote
        # (nothing of the library touches them in between), so its device state is taken over without comparing them
        trusted = self.__dict__.pop('_imp_trusted', None) is self.imp
        try:
            later = early and not refresh    # (a refresh reads the numpy inputs: no deferred detach then)
            if i == 1:
                with self._init_scale():
                    self.imp.sample(burnin=ess_burn, detach=not later, trusted=trusted)
            else:
                self.imp.sample(burnin=ess_burn, detach=not later, trusted=trusted)
            if refresh:
                self.imp.update_ord_nn()
            if early:
                self._m_step(early=True)
            else:
                self._m_step()
        finally:
            self.imp.finish_detach()   # (whatever happened: the nodes' numpy attributes are the state the sampler left)
        self._imp_trusted = self.imp

    def _mstep_can_start_early(self):
        """True when the M-step's first objective evaluations can be queued on the device BEFORE the host has refreshed the
        nodes' numpy attributes from the I-step's device state (imputer.sample(detach=False)): every GP node is fitted by a
        lock-step driver (all Vecchia or all dense), from the imputer's own device views -- no reference prior (compute_cl reads the
        numpy input), no replicates, one rank."""
        from . import mstep
        if not mstep._HAVE_CORE or ddist.is_active() or os.environ.get('DGPAMD_MSTEP_EARLY', '1') == '0':
            return False
        gps = [nd for layer in self.all_layer for nd in layer if nd.type == 'gp']
        if len(gps) < 2:
            return False
        if not self.vecch:
            # Dense models (round 6): every GP node is fitted by the dense lock-step driver from the imputer's device views; its first round is queued
            # (dgpamd_llik_batch_launch) before the host refreshes the numpy attributes and runs the R2 diagnostics -- 0.6 ms of idle device per iteration
            # at the bench shape (profiles/r06_idle_gaps.txt).  Only GP nodes without replicates or a reference prior, as below.
            for layer in self.all_layer:
                for nd in layer:
                    if nd.type != 'gp' or nd.vecch or nd.rep is not None or not isinstance(nd, TrackedInputs) or nd.prior_name == 'ref':
                        return False
            return True
        for layer in self.all_layer:
            for nd in layer:
                if nd.rep is not None or not isinstance(nd, TrackedInputs):
                    return False
                if nd.type == 'gp' and (not nd.vecch or nd.prior_name == 'ref' or (nd.target == 'gp' and len(nd.length) != 1)):
                    return False
        return True

    def _handoff_fallback(self, exc, marks=()):
        """Undo what the interrupted iteration committed (nodes fitted before the failure have appended a row to their
        para_path -- estimate()'s burn-in index and reinit_all_layer(row=...) count rows -- and the lock-step M-step may have
        handed factors / log-likelihoods of its last evaluations to the imputer), then switch to the per-block-step kernel.
        (A neighbour refresh of the interrupted attempt has drawn its permutation; the repeat draws another -- any ordering is
        a valid one.)"""
        import warnings
        for nd, rows in marks:
            if rows is not None and nd.para_path is not None and len(nd.para_path) > rows:
                nd.para_path = nd.para_path[:rows]
        warnings.warn('dgp_amd: %s -- this engine now factors with one launch per block step (set_potrf_mode(0)); the '
                      'iteration is repeated' % exc, RuntimeWarning)
        self.engine.sync()
        self.engine.set_potrf_mode(0)
        imp = getattr(self, 'imp', None)
        if imp is not None:   # nothing computed by the interrupted launches may be reused
            imp._ll_cache, imp._factor_cache = {}, {}
            for key in ('_want_ll0', '_adopt', '_adopt_ll'):
                imp.__dict__.pop(key, None)

    def ptrain(self, N=500, ess_burn=10, disable=False, core_num=None):
        """dgp.py:1414-1472 optimised the nodes of a layer in a process pool; here they already run concurrently on the
        device (`core_num` is accepted and unused)."""
        return self.train(N=N, ess_burn=ess_burn, disable=disable)


    def _set_final_output(self, nd, k):
        """Observed output of final-layer node k from self.Y (site means / weights with replicates, dgp.py:1344-1356)."""
        if nd.type == 'likelihood' or nd.rep is None:
            nd.output = self.Y[:, [k]].copy()
            return
        G = nd.rep.max() + 1
        nd.W_diag = 1.0 / np.bincount(nd.rep, minlength=G)
        nd.output = (np.bincount(nd.rep, weights=self.Y[:, k], minlength=G) * nd.W_diag).reshape(-1, 1)
        res = self.Y[:, [k]] - nd.output[nd.rep, :]
        nd.sum_residual = (res.T @ res).flatten()

    def reinit_all_layer(self, reset_lengthscale, row=0, truncate=True):
        """Fresh latent warm start from the current (X, Y), hyper-parameters back to para_path[row] if asked
        (dgp.py:1097-1362, GP nodes and Hetero).  truncate: drop the path after `row` (a restarted train() call must not
        leave the failed attempt's iterations in para_path; update_xy keeps the history like the reference)."""
        global_in = In = self.X
        for l, layer in enumerate(self.all_layer):
            last = l == self.n_layer - 1
            Out = None if last else self._layer_warm_start(l, In, len(layer))
            for k, nd in enumerate(layer):
                if last:
                    nd.rep = self.indices
                if nd.type == 'likelihood':
                    bind_private(nd, 'input', In[nd.rep, :][:, nd.input_dim] if nd.rep is not None else In[:, nd.input_dim])
                    nd.output = self.Y[:, [k]].copy()
                    continue
                bind_private(nd, 'input', In[:, nd.input_dim])
                if nd.type == 'gp':
                    if nd.connect is not None:
                        bind_private(nd, 'global_input', global_in[:, nd.connect])
                    nd.m = self.m
                    if reset_lengthscale:
                        est = nd.para_path[row]
                        nd.scale, nd.length, nd.nugget = np.atleast_1d(est[0]), np.atleast_1d(est[1:-1]), np.atleast_1d(est[-1])
                        if truncate:
                            nd.para_path = nd.para_path[:row + 1]
                if last:
                    self._set_final_output(nd, k)
                else:
                    nd.output = Out[:, [k]].copy()
                if nd.type == 'gp' and nd.prior_name == 'ref':
                    nd.compute_cl()
            if self.vecch:
                self._layer_ord_nn(layer)
            if not last:
                In = copy.copy(Out)
        stats = self.imp.stats
        self.imp = imputer(self.all_layer, self.block, draws=self.draws, engine=self.engine)   # (the data may have changed size)
        self.imp.stats = stats
        self.imp.sample(burnin=10)
        self.compute_r2()

    def update_all_layer(self, all_layer):
        """Replace the structure by one that carries hyper-parameter and latent values already, e.g. `estimate()` of
        another run (dgp.py:760-822): the parameter paths restart there, 10 sweeps of the sampler, N = 0."""
        self.all_layer = all_layer
        self.n_layer = len(all_layer)
        for l, layer in enumerate(all_layer):
            for nd in layer:
                if l == self.n_layer - 1 and nd.rep is not None:
                    self.indices = nd.rep
                if nd.type != 'gp':
                    continue
                nd.engine = self.engine
                nd.para_path = np.atleast_2d(np.concatenate((nd.scale, nd.length, nd.nugget)))
                nd.D = peek(nd, 'input').shape[1] + (0 if nd.connect is None else len(nd.connect))
                if nd.prior_name == 'ref':
                    p = nd.D
                    nd.prior_coef[1] = 1 / len(nd.output) ** (1 / p) * (nd.prior_coef[0] + p)
                    nd.compute_cl()
            if self.vecch:
                self._layer_ord_nn(layer)
        self.imp = imputer(self.all_layer, self.block, draws=self.draws, engine=self.engine)
        self.imp.sample(burnin=10)
        self.compute_r2()
        self.N = 0
        self.burnin = None

    # ------------------------------------------------------------------ new data (sequential design)
    def update_xy(self, X, Y, reset=False):
        """Replace the training data of a trained DGP (dgp.py:824-888).  reset=True: latents and hyper-parameters start
        over.  Otherwise the structure is warm started: if the new inputs are a subset of the old ones the latents are
        subsetted; if they are a superset the latents at the new sites are the nodes' conditional means given the
        current latents, layer by layer; else a fresh warm start with the current hyper-parameters."""
        if isinstance(Y, list):
            if len(Y) != 1:
                raise Exception('Y has to be a numpy 2d-array rather than a list. The list version of Y (for linked '
                                'emulation) has been reduced. Please use the dedicated lgp class for linked emulation.')
            Y = Y[0]
        if Y.ndim == 1 or X.ndim == 1:
            raise Exception('The input and output data have to be numpy 2d-arrays.')
        self.Y = Y
        self._encode_labels(fit=False)
        self.indices = self.counts = None
        origin_X = self.X.copy()
        self.X = X
        if self.check_rep:
            X0, inv, counts = np.unique(X, return_inverse=True, return_counts=True, axis=0)
            if len(X0) != len(X):
                self.X, self.indices, self.counts = X0, np.asarray(inv).reshape(-1), counts
        self.n_data = self.X.shape[0]
        self.m = min(self.m, self.n_data - 1)
        if reset:
            self.reinit_all_layer(reset_lengthscale=True, truncate=False)
        else:
            new_in_old = (self.X[:, None] == origin_X).all(-1)
            if new_in_old.any(-1).all():
                self._update_all_layer_smaller(np.where(new_in_old)[1])
                burn = 50
            elif new_in_old.any(0).all():
                self._update_all_layer_larger(np.where((self.X == origin_X[:, None]).all(-1))[1])
                burn = 50
            else:
                self.reinit_all_layer(reset_lengthscale=False)
                burn = 190      # (reinit_all_layer already ran 10 sweeps: 200 in total as the reference)
            self.imp = imputer(self.all_layer, self.block, draws=self.draws, engine=self.engine)
            self.imp.sample(burnin=burn)
            self.compute_r2()

    def _update_all_layer_larger(self, sub_idx):
        """The old inputs are rows sub_idx of the new ones (dgp.py:890-1012): hidden latents at the new sites = each
        node's GP conditional mean given its current input/output pairs, propagated layer by layer."""
        global_in = In = self.X
        mask = np.zeros(len(self.X), dtype=bool)
        mask[sub_idx] = True
        for l, layer in enumerate(self.all_layer):
            last = l == self.n_layer - 1
            Out = None if last else np.empty((len(In), len(layer)))
            for k, nd in enumerate(layer):
                if nd.type == 'gp':
                    nd.m = self.m
                if not last:
                    zz = None if nd.connect is None else global_in[~mask, :][:, nd.connect]
                    if nd.vecch:
                        nd.pred_m = 50
                    else:
                        nd.compute_stats()
                    mu = nd.gp_prediction(In[~mask, :][:, nd.input_dim], zz)[0]
                    Out[sub_idx, k] = nd.output.flatten()
                    Out[~mask, k] = np.asarray(mu).flatten()
                    bind_private(nd, 'input', In[:, nd.input_dim].copy())
                    nd.output = Out[:, [k]].copy()
                    if nd.connect is not None:
                        bind_private(nd, 'global_input', global_in[:, nd.connect].copy())
                    nd._stats = None
                else:
                    nd.rep = self.indices
                    if nd.type == 'likelihood' and nd.rep is not None:
                        bind_private(nd, 'input', In[nd.rep, :][:, nd.input_dim].copy())
                    else:
                        bind_private(nd, 'input', In[:, nd.input_dim].copy())
                    if nd.type == 'gp' and nd.connect is not None:
                        bind_private(nd, 'global_input', global_in[:, nd.connect].copy())
                    self._set_final_output(nd, k)
                if nd.type == 'gp' and nd.prior_name == 'ref':
                    nd.compute_cl()
            if self.vecch:
                self._layer_ord_nn(layer)
            if not last:
                In = Out.copy()

    def _update_all_layer_smaller(self, sub_idx):
        """The new inputs are rows sub_idx of the old ones (dgp.py:1014-1095): latents are subsetted."""
        for l, layer in enumerate(self.all_layer):
            last = l == self.n_layer - 1
            for k, nd in enumerate(layer):
                if last and nd.type == 'likelihood':
                    if nd.rep is not None:    # back to one row per distinct site first
                        bind_private(nd, 'input', np.concatenate([np.unique(peek(nd, 'input')[nd.rep == i, :], axis=0) for i in range(np.max(nd.rep) + 1)], axis=0))
                    bind_private(nd, 'input', peek(nd, 'input')[sub_idx, :])
                    if self.indices is not None:
                        bind_private(nd, 'input', peek(nd, 'input')[self.indices, :])
                else:
                    bind_private(nd, 'input', peek(nd, 'input')[sub_idx, :])
                if last:
                    nd.rep = self.indices
                if nd.type == 'gp':
                    if nd.connect is not None:
                        bind_private(nd, 'global_input', self.X[:, nd.connect].copy())
                    nd.m = self.m
                    nd._stats = None
                if last:
                    self._set_final_output(nd, k)
                else:
                    nd.output = nd.output[sub_idx, :].copy()
                if nd.type == 'gp' and nd.prior_name == 'ref':
                    nd.compute_cl()
            if self.vecch:
                self._layer_ord_nn(layer)

    update_all_layer_larger = _update_all_layer_larger      # the reference's names (dgp.py:933,1014)
    update_all_layer_smaller = _update_all_layer_smaller

    def plot(self, layer_no, ker_no, width=4., height=1., ticksize=5., labelsize=8., hspace=0.1):
        """Trace plots of the parameters (variance, lengthscales, nugget) of GP node ker_no of layer layer_no, both
        counted from one (dgp.py:1543-1572)."""
        nd = self.all_layer[layer_no - 1][ker_no - 1]
        if nd.type != 'gp':
            print('There is nothing to plot for a likelihood node, please choose a GP node instead.')
            return
        import matplotlib.pyplot as plt
        n_para = nd.para_path.shape[1]
        labels = [r'$\sigma^2$'] + [r'$\gamma_{%i}$' % p for p in range(1, n_para - 1)] + [r'$\eta$']
        fig, axes = plt.subplots(n_para, figsize=(width, n_para * height), dpi=100, sharex=True)
        fig.tight_layout()
        fig.subplots_adjust(hspace=hspace)
        for ax, trace, lab in zip(np.atleast_1d(axes), nd.para_path.T, labels):
            ax.plot(trace)
            ax.tick_params(axis='both', which='major', labelsize=ticksize)
            ax.set_ylabel(lab, fontsize=labelsize)
        plt.show()

    def change_init_scale(self):
        """dgp.py:1575-1586 under its reference name."""
        return self._init_scale()

    def estimate(self, burnin=None):
        """Point estimates = mean of para_path[burnin:], burnin default int(0.75 N) (dgp.py:1517-1541)."""
        self.burnin = int(self.N * (3 / 4)) if burnin is None else burnin
        final = copy.deepcopy(self.all_layer)
        for layer in final:
            for nd in layer:
                if nd.type == 'gp':
                    est = np.mean(nd.para_path[self.burnin:, :], axis=0)
                    nd.scale, nd.length, nd.nugget = np.atleast_1d(est[0]), np.atleast_1d(est[1:-1]), np.atleast_1d(est[-1])
        return final

    def aggregate_r2(self, burnin=0.75, agg='median'):
        if burnin < 0 or burnin > 1:
            raise Exception('burnin must be between 0 and 1.')
        f = {'mean': np.mean, 'median': np.median}.get(agg)
        if f is None:
            raise Exception("agg must be either 'median' or 'mean'.")
        return [[None if (nd.type != 'gp' or nd.R2 is None) else f(nd.R2[int(len(nd.R2) * burnin):, :], axis=0)
                 for nd in layer] for layer in self.all_layer]
