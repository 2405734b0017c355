"""Prediction from a trained DGP by stochastic imputation -- mirror of dgpsi.emulator
(emulation.py:24-44 construction, :631-854 predict(method='mean_var')).

Construction draws the N imputations from one ESS chain after a burn-in
(emulation.py:31-44).  Instead of deep-copying the whole structure with an n x n
R^-1 (and D n x n Psexp) per node per imputation, the emulator keeps per
imputation only the latent columns and builds device-resident statistics:
  * first-layer nodes: inputs and hyper-parameters are identical in every
    imputation, so R^-1 is factored ONCE and the N vectors R^-1 y_s ride along as
    right-hand sides of that single factorisation; the predictive variance is
    shared, only the mean differs;
  * deeper nodes: one factorisation per imputation (their inputs differ).
`predict` walks the layers on the device and accumulates the imputation moments
mu = mean_s mu_s, var = mean_s(mu_s^2 + v_s) - mu^2 (emulation.py:846-847) in place.

Multi-GPU: with torch.distributed initialised, each rank draws its share of the N
imputations from its own chain (own burn-in) and one all-reduce(sum) of the two
moment arrays precedes the finalisation (RCCL over xGMI on the GPU box).
"""
import contextlib
import copy

import collections
import os

import numpy as np
from .kernel_class import bind_private, peek
import torch

from .imputation import imputer, DrawStream
from .ops import default_engine
from . import dist as ddist


class _LazyPer:
    """The per-imputation prediction statistics of a linked GP node (R^-1, R^-1 y, its inputs: kernel_class.py:735-764 for
    every imputation, emulation.py:43), built when first asked for and kept while the emulator's byte budget allows.
    The reference holds all of them; at BASELINE configs[2]'s size -- 50 imputations x 3 nodes, n = 5000 -- that is 31 GB of
    inverses of which a predict() call uses each exactly once, and rebuilding one costs ~30 ms against the seconds its pair
    kernel runs.  Budget: DGPAMD_STATS_GB (default 16); smaller models never evict."""

    def __init__(self, owner, key, build, nbytes):
        self.owner, self.key, self.build, self.nbytes = owner, key, build, int(nbytes)

    def __getitem__(self, s):
        o = self.owner
        ck = self.key + (int(s),)
        hit = o._per_cache.get(ck)
        if hit is not None:
            o._per_cache.move_to_end(ck)
            return hit
        budget = float(os.environ.get('DGPAMD_STATS_GB', '16')) * 2 ** 30
        while o._per_cache and o._per_bytes + self.nbytes > budget:
            _, old = o._per_cache.popitem(last=False)
            o._per_bytes -= old['_bytes']
        item = self.build(int(s))
        item['_bytes'] = self.nbytes
        o._per_cache[ck] = item
        o._per_bytes += self.nbytes
        return item


class emulator:
    """Args as dgpsi.emulator (emulation.py:24): all_layer (from dgp.estimate()), N, block;
    plus `seed`, `device` and `shard`: None / True / False -- the N imputations are split over the ranks of an
    initialised torch.distributed group (None: iff there is one) and predictions cost one all-reduce of the two moment
    sums; 'points' -- every rank holds all N imputations (same seed, same chain) and predict() splits the rows of x over
    the ranks instead, one all-gather of the results (what ppredict's pool did, emulation.py:578-629)."""

    def __init__(self, all_layer, N=10, block=True, seed=None, device=None, shard=None):
        self.all_layer = all_layer
        self.n_layer = len(all_layer)
        self.vecch = bool(all_layer[0][0].vecch)
        self.N_total = int(N)
        self.shard_points = shard == 'points' and ddist.is_active()
        self.shard = False if shard == 'points' else (ddist.is_active() if shard is None else bool(shard))
        rank, world = (ddist.rank(), ddist.world()) if self.shard else (0, 1)
        if self.shard and self.N_total < world:   # (a rank without imputations would skip the collectives the others enter)
            raise Exception('emulator(shard=True) needs at least one imputation per rank: N = %d < %d ranks' % (self.N_total, world))
        self.engine = default_engine(device)
        for layer in all_layer:
            for nd in layer:
                if nd.type == 'gp':
                    nd.engine = self.engine
        self.N = ddist.share(self.N_total, rank, world)
        if self.shard_points and seed is None:      # all ranks must draw the same imputations: rank 0's entropy for everyone
            seed = ddist.broadcast_int(np.random.SeedSequence().entropy, src=0, device=self.engine.device)   # (no pickling: RCCL moves device words)
        ss = np.random.SeedSequence(seed)
        self.imp = imputer(all_layer, block, draws=DrawStream(ss.spawn(world)[rank]), engine=self.engine)
        self._sample_rng = np.random.default_rng(ss.spawn(world)[rank])   # predict(method='sampling')
        if self.vecch:
            self.imp.update_ord_nn()
            self.imp.sample(burnin=20)
        else:
            self.imp.sample(burnin=50)
        # per imputation: the latent columns of every hidden layer (n x M_l), small
        self.latents = []
        self.orders = []
        for _ in range(self.N):
            if self.vecch:
                self.imp.update_ord_nn()
            self.imp.sample()
            self.latents.append([np.stack([np.asarray(nd.output, float).reshape(-1) for nd in layer], 1)
                                 for layer in all_layer[:-1]])
            if self.vecch:
                self.orders.append([[(nd.ord.copy(), nd.NNarray.copy()) if nd.type == 'gp' else None for nd in layer]
                                    for layer in all_layer])
        self._stats = None

    def to_vecchia(self):
        """Switch the emulator to Vecchia predictions (emulation.py:63-74): every node conditions on its nearest training
        points instead of using stored n x n statistics."""
        if self.vecch:
            raise Exception('The DGP emulator is already in Vecchia mode.')
        self.vecch = True
        for layer in self.all_layer:
            for nd in layer:
                if nd.type == 'gp':
                    nd.vecch = True
        self._stats = None

    def remove_vecchia(self):
        """Back to dense predictions (emulation.py:76-88); the statistics are rebuilt on the next predict()."""
        if not self.vecch:
            raise Exception('The DGP emulator is already in non-Vecchia mode.')
        self.vecch = False
        for layer in self.all_layer:
            for nd in layer:
                if nd.type == 'gp':
                    nd.vecch = False
        self._stats = None

    def __getstate__(self):
        st = dict(self.__dict__)
        st['engine'] = None       # device context and statistics are rebuilt after unpickling (utils.write / read)
        st['_stats'] = None
        st.pop('_per_cache', None)
        st.pop('_per_bytes', None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.engine = default_engine()
        for layer in self.all_layer:
            for nd in layer:
                if nd.type == 'gp':
                    nd.engine = self.engine
        self.imp._engine = self.engine

    # dgpsi keeps `all_layer_set`: N deep copies of the structure.  Built on demand (arrays only, no R^-1).
    @property
    def all_layer_set(self):
        out = []
        for s in range(self.N):
            out.append(self._structure(s))
        return out

    def _structure(self, s):
        al = copy.deepcopy(self.all_layer)
        lat = self.latents[s]
        for l, layer in enumerate(al):
            for k, nd in enumerate(layer):
                if l < self.n_layer - 1:
                    nd.output = lat[l][:, [k]].copy()
                if l > 0:
                    bind_private(nd, 'input', lat[l - 1][:, nd.input_dim].copy())
        return al

    # ------------------------------------------------------------------ statistics
    def _build_stats(self):
        """R^-1 (device, ld = Np) and R^-1 y for every GP node and imputation."""
        e = self.engine
        S = self.N
        stats = {}
        Yall = np.asarray([np.asarray(nd.output, float).reshape(-1) for nd in self.all_layer[-1]])
        for l, layer in enumerate(self.all_layer):
            for k, nd in enumerate(layer):
                if nd.type != 'gp':
                    continue
                n = len(nd.output)
                Np = e.padded_dim(n)
                cap = Np - n
                Xg = None if peek(nd, 'global_input') is None else e.tensor(peek(nd, 'global_input'))
                W = None if nd.rep is None else e.tensor(nd.W_diag)

                def ys(s, l=l, k=k):   # (defaults: the closures below are called after this loop has moved on)
                    return self.latents[s][l][:, k] if l < self.n_layer - 1 else Yall[k]

                def factor(Xl, Y, nd=nd, n=n, Np=Np, Xg=Xg, W=W):
                    A = e.workspace(('emuA', n), Np * Np * 8)
                    Ainv = e.empty(Np, Np)
                    e.kmatrix(nd.name, Xl, None, Xg, nd.length, nd.nugget[0], W=W, out=A, full=False, Y=Y)
                    work = e.potrf_workspace(n, 1)
                    _, info = e.potrf(n, A, work=work)
                    r = Y.shape[0]
                    e.potri(n, A, Ainv, r, work)
                    if int(info.cpu().numpy()[0]):   # not numerically PD: pseudo-inverse (kernel_class.py:749-751)
                        K = e.kmatrix(nd.name, Xl, None, Xg, nd.length, nd.nugget[0], W=W, full=True)
                        Ainv.zero_()
                        Ainv[:n, :n] = e.pinvh(K)
                        return Ainv, (Y @ Ainv[:n, :n]).contiguous()
                    return Ainv, (-Ainv[n:n + r, :n]).contiguous()

                if l == 0:
                    Xl = e.tensor(peek(nd, 'input'))
                    rys, Rinv = [], None
                    for c0 in range(0, S, cap):   # all imputations' y as right-hand sides of ONE factorisation
                        Y = e.tensor(np.stack([ys(s) for s in range(c0, min(S, c0 + cap))]))
                        Rinv, ry = factor(Xl, Y)
                        rys.append(ry)
                    stats[(l, k)] = dict(shared=True, Rinv=Rinv, ld=Np, ry=torch.cat(rys), n=n, Wall=e.tensor(nd._X()))
                else:
                    def build(s, l=l, nd=nd, factor=factor, ys=ys, Xg=Xg):
                        Xin = self.latents[s][l - 1][:, nd.input_dim]
                        Rinv, ry = factor(e.tensor(Xin), e.tensor(ys(s)[None, :]))
                        cells = e.linkgp_cells(nd.name, Xin, Xg, Rinv, ry[0])   # (Matern: training points grouped by cells)
                        return cells if cells is not None else dict(Rinv=Rinv, ry=ry[0].contiguous(), W=e.tensor(Xin))
                    stats[(l, k)] = dict(shared=False, per=_LazyPer(self, (l, k), build, Np * Np * 8), ld=Np, n=n, Wg=Xg)
        self._stats = stats
        self._per_cache = collections.OrderedDict()
        self._per_bytes = 0

    def _layer_moments(self, x):
        """Per layer the (mean, variance) of every node at the rows of x for every imputation held by this rank, as
        device tensors (S, M, K) -- the layer walk of emulation.py:701-779 (dense mode)."""
        if self._stats is None:
            self._build_stats()
        e = self.engine
        M, S = len(x), self.N
        xd = e.tensor(x)
        per_layer = []
        for l, layer in enumerate(self.all_layer):
            if l == self.n_layer - 1 and self._cat() is not None:
                idx = torch.as_tensor(np.asarray(self._cat().input_dim), device=xd.device)
                per_layer.append((per_layer[-1][0][:, :, idx].contiguous(), per_layer[-1][1][:, :, idx].contiguous()))
                continue
            K = len(layer)
            mean = e.empty(S, M, K)
            var = e.empty(S, M, K)
            for k, nd in enumerate(layer):
                if nd.type != 'gp':   # likelihood node: closed-form moments of y from the feeding latents' (host protocol)
                    pm, pv = (t.cpu().numpy() for t in per_layer[-1])
                    for s in range(S):
                        mk, vk = nd.prediction(m=pm[s][:, nd.input_dim], v=pv[s][:, nd.input_dim])
                        mean[s, :, k] = e.tensor(mk)
                        var[s, :, k] = e.tensor(vk)
                    continue
                st = self._stats[(l, k)]
                z = None if nd.connect is None else xd[:, torch.as_tensor(nd.connect, device=xd.device)].contiguous()
                if l == 0:
                    xin = xd[:, torch.as_tensor(nd.input_dim, device=xd.device)]
                    xin = (xin if z is None else torch.cat((xin, z), 1)).contiguous()
                    mk, vk = e.gp_predict(nd.name, xin, st['Wall'], nd.length, st['Rinv'], st['ld'], st['ry'], nd.scale[0],
                                          nd.nugget[0])
                    mean[:, :, k] = mk
                    var[:, :, k] = vk[None, :]
                else:
                    pm, pv = per_layer[-1]
                    idx = torch.as_tensor(nd.input_dim, device=xd.device)
                    for s in range(S):
                        ps = st['per'][s]
                        mk, vk = e.linkgp_predict(nd.name, pm[s][:, idx].contiguous(), pv[s][:, idx].contiguous(), z, ps['W'],
                                                  ps.get('Wg', st['Wg']), nd.length, ps['Rinv'], st['ld'], ps['ry'], nd.scale[0],
                                                  nd.nugget[0])
                        mean[s, :, k] = mk
                        var[s, :, k] = vk
            per_layer.append((mean, var))
        return per_layer

    def _layer_moments_loo(self, x):
        """Leave-one-out layer walk of a dense emulator (emulation.py:90-143 with vecch False: every node conditions
        on all training points but its own -- test row k leaves out training row k at every layer, which is what
        get_pred_nn's all-points shortcut (vecchia.py:23-26) followed by kernel_class.py:610-611,655-656 does).  Nothing is
        refactorised: first-layer nodes use the block-inverse identities  mean = y_d - (R^-1 y)_d / (R^-1)_dd,
        var = scale (1 / (R^-1)_dd + nugget (1 - W_d));  linked nodes go through dgpamd_linkgp_loo, which applies the
        rank-one downdate of R^-1 inside the pair weights.  Returns device (S, M, K) pairs per layer."""
        if self._stats is None:
            self._build_stats()
        e = self.engine
        M, S = len(x), self.N
        xd = e.tensor(x)
        per_layer = []
        for l, layer in enumerate(self.all_layer):
            if l == self.n_layer - 1 and self._cat() is not None:
                idx = torch.as_tensor(np.asarray(self._cat().input_dim), device=xd.device)
                per_layer.append((per_layer[-1][0][:, :, idx].contiguous(), per_layer[-1][1][:, :, idx].contiguous()))
                continue
            K = len(layer)
            mean, var = e.empty(S, M, K), e.empty(S, M, K)
            for k, nd in enumerate(layer):
                if nd.type != 'gp':
                    pm, pv = (t.cpu().numpy() for t in per_layer[-1])
                    for s in range(S):
                        mk, vk = nd.prediction(m=pm[s][:, nd.input_dim], v=pv[s][:, nd.input_dim])
                        mean[s, :, k], var[s, :, k] = e.tensor(mk), e.tensor(vk)
                    continue
                st = self._stats[(l, k)]
                n = st['n']
                z = None if nd.connect is None else xd[:, torch.as_tensor(nd.connect, device=xd.device)].contiguous()
                if l == 0:
                    xin = xd[:, torch.as_tensor(nd.input_dim, device=xd.device)]
                    xin = (xin if z is None else torch.cat((xin, z), 1)).contiguous()
                    if M != n or not torch.equal(st['Wall'], xin):
                        raise Exception('loo: the rows of X must be the training input positions of the emulator, in order.')
                    d = torch.arange(n, device=xd.device)
                    rho = st['Rinv'][:n, :n].diagonal()[d]
                    wd = 1.0 if nd.rep is None else e.tensor(nd.W_diag)[d]
                    for s in range(S):
                        y = e.tensor(self.latents[s][l][:, k] if l < self.n_layer - 1 else
                                     np.asarray(nd.output, float).reshape(-1))
                        mean[s, :, k] = y[d] - st['ry'][s][d] / rho
                    var[:, :, k] = (nd.scale[0] * (1.0 / rho + nd.nugget[0] * (1.0 - wd)))[None, :]
                else:
                    pm, pv = per_layer[-1]
                    idx = torch.as_tensor(nd.input_dim, device=xd.device)
                    for s in range(S):
                        ps = st['per'][s]
                        ms, vs = pm[s][:, idx].contiguous(), pv[s][:, idx].contiguous()
                        d = ps['pos'] if 'pos' in ps else torch.arange(n, device=xd.device, dtype=torch.int32)
                        mk, vk = e.linkgp_predict(nd.name, ms, vs, z, ps['W'], ps.get('Wg', st['Wg']), nd.length, ps['Rinv'], st['ld'],
                                                  ps['ry'], nd.scale[0], nd.nugget[0], drop=d)
                        mean[s, :, k], var[s, :, k] = mk, vk
            per_layer.append((mean, var))
        return per_layer

    def _layer_moments_vecchia(self, x, m):
        """The same layer walk in Vecchia mode (no stored statistics; every node conditions on its pred_m nearest
        neighbours, kernel_class.py:603-619,647-664).  Returns numpy (S, M, K) pairs per layer."""
        M, S = len(x), self.N
        layers = [[] for _ in self.all_layer]
        # Conditioning sets are searched once per GROUP of nodes that must get the same ones: nodes of a layer with the same
        # input columns and ONE shared lengthscale see the same points in the same (distance, index) order whatever the
        # lengthscale's value (a uniform scaling; the reference itself shares orderings between such siblings in training,
        # imputation.py:245-262) -- and in the first layer also across imputations, whose inputs are the same X.  At
        # BASELINE configs[3] (8 + 1 nodes, 2 imputations) that is 3 searches instead of 18, which were 60 % of a large
        # Vecchia prediction.  DGPAMD_NN_SHARE=0 searches per node like the reference (vecchia.py:20-40 per gp_prediction).
        share = os.environ.get('DGPAMD_NN_SHARE', '1') != '0'
        nn_sets = {}

        def hand_over(nd, l, s_, xq):
            if not share or nd.loo_state:
                return
            iso = len(nd.length) == 1
            key = (l, None if l == 0 else s_, tuple(np.asarray(nd.input_dim).tolist()),
                   None if nd.connect is None else tuple(np.asarray(nd.connect).tolist()), 'iso' if iso else tuple(nd.length.tolist()), m)
            if key not in nn_sets:
                nn_sets[key] = nd._pred_nn(xq, nd._X())
            nd._nn_given = nn_sets[key]

        for s in range(S):
            al = self._structure(s)
            m_in = v_in = None
            for l, layer in enumerate(al):
                if l == self.n_layer - 1 and self._cat() is not None:
                    idx = np.asarray(layer[0].input_dim)
                    layers[l].append((m_in[:, idx].copy(), v_in[:, idx].copy()))
                    continue
                mo, vo = np.empty((M, len(layer))), np.empty((M, len(layer)))
                for k, nd in enumerate(layer):
                    if nd.type == 'gp':
                        nd.engine = self.engine
                        nd.pred_m = m
                    if nd.type != 'gp':
                        mo[:, k], vo[:, k] = nd.prediction(m=m_in[:, nd.input_dim], v=v_in[:, nd.input_dim])
                        continue
                    z = None if nd.connect is None else x[:, nd.connect]
                    try:
                        if l == 0:
                            xq = x[:, nd.input_dim]
                            hand_over(nd, l, s, xq if z is None else np.concatenate((xq, z), 1))
                            mo[:, k], vo[:, k] = nd.gp_prediction(xq, z)
                        else:
                            mq = m_in[:, nd.input_dim]
                            hand_over(nd, l, s, mq if z is None else np.concatenate((mq, z), 1))
                            mo[:, k], vo[:, k] = nd.linkgp_prediction(mq, v_in[:, nd.input_dim], z)
                    finally:
                        # a set this call did not consume (a branch without a neighbour search, an exception on the way)
                        # must not meet a later prediction with the same number of rows
                        nd.__dict__.pop('_nn_given', None)
                m_in, v_in = mo, vo
                layers[l].append((mo, vo))
        return [(np.stack([a for a, _ in L]), np.stack([b for _, b in L])) for L in layers]

    def _cat(self):
        """The Categorical likelihood node of the final layer, or None.  For it the last layer's moments are those of
        its feeding latents (emulation.py:711-716,751-752): they are aggregated over the imputations first and turned
        into class probabilities afterwards."""
        nd = self.all_layer[-1][0]
        return nd if getattr(nd, 'name', None) == 'Categorical' else None

    # ------------------------------------------------------------------ prediction
    def predict(self, x, method='mean_var', full_layer=False, sample_size=50, m=50, aggregation=True):
        """Mean and variance at the rows of x (emulation.py:631-854, method='mean_var').
        Returns (mu, sigma2) as numpy arrays (M x D_out), or per-layer lists if full_layer, or the
        per-imputation lists if aggregation=False."""
        if x.ndim == 1:
            raise Exception('The testing input has to be a numpy 2d-array')
        if method not in ('mean_var', 'sampling'):
            raise Exception("method must be either 'mean_var' or 'sampling'.")
        if getattr(self, 'shard_points', False) and not getattr(self, '_in_points', False):
            return self._predict_points(x, method, full_layer, sample_size, m, aggregation)
        if self.vecch:
            return self._predict_vecchia(x, full_layer, m, aggregation, method, sample_size)
        if self.shard and (method == 'sampling' or not aggregation):
            # (each rank holds its own imputations only: the per-imputation lists / draws would silently be partial)
            raise NotImplementedError("with the imputations sharded over ranks predict() returns aggregated moments only; "
                                      "use emulator(..., shard=False) or shard='points' for method='sampling' / aggregation=False")
        e = self.engine
        M, S = len(x), self.N
        per_layer = self._layer_moments(x)
        if method == 'sampling':
            return self._draw_samples([(mean.cpu().numpy(), var.cpu().numpy()) for mean, var in per_layer], sample_size,
                                      full_layer)
        cat = self._cat()
        if not aggregation and not full_layer:
            mu_s, v_s = per_layer[-1]
            mu_s, v_s = [t.cpu().numpy() for t in mu_s], [t.cpu().numpy() for t in v_s]
            if cat is not None:
                pr = [cat.prediction(a, b) for a, b in zip(mu_s, v_s)]
                return [p[0] for p in pr], [p[1] for p in pr]
            return mu_s, v_s
        outs = []
        for mean, var in (per_layer if full_layer else per_layer[-1:]):
            s1, s2 = e.zeros(M, mean.shape[2]), e.zeros(M, mean.shape[2])
            for s in range(S):
                e.moments_accumulate(mean[s].contiguous(), var[s].contiguous(), s1, s2)
            if self.shard:
                ddist.allreduce_sum(s1, s2)
            e.moments_finalize(self.N_total if self.shard else S, s1, s2)
            outs.append((s1.cpu().numpy(), s2.cpu().numpy()))
        if cat is not None:
            outs[-1] = cat.prediction(outs[-1][0], outs[-1][1])
        if full_layer:
            return [o[0] for o in outs], [o[1] for o in outs]
        return outs[0]

    def _draw_samples(self, per_layer, sample_size, full_layer):
        """method='sampling' (emulation.py:780-822, GP hierarchies): per imputation and layer the outputs are drawn
        from N(mu_s, sigma2_s), sample_size times.  per_layer: [(mean (S,M,K), var (S,M,K))] as numpy arrays.
        Returns, like the reference, a list over the final layer's nodes of (M x S*sample_size) arrays, or with
        full_layer a list over layers of such lists."""
        rng = self._sample_rng
        lik = any(nd.type == 'likelihood' for nd in self.all_layer[-1])
        out, prev = [], None
        for li, (mean, var) in enumerate(per_layer):
            last = li == len(per_layer) - 1
            if not (full_layer or last or (lik and li == len(per_layer) - 2)):
                continue
            S, M, K = mean.shape
            mu_r, sd_r = np.repeat(mean, sample_size, axis=0), np.repeat(np.sqrt(var), sample_size, axis=0)
            draws = rng.normal(mu_r, sd_r)                       # (S*ss, M, K)
            if last and self._cat() is not None:   # class probabilities at draws of the feeding latents (all columns at once)
                cat = self._cat()
                draws = np.stack([cat.sampling(prev[j][:, cat.input_dim]) for j in range(prev.shape[0])])
            elif last and lik:   # likelihood nodes sample y from draws of their feeding latents (emulation.py:785-822)
                for k, nd in enumerate(self.all_layer[-1]):
                    if nd.type == 'likelihood':
                        for j in range(draws.shape[0]):
                            draws[j, :, k] = nd.sampling(prev[j][:, nd.input_dim])
            prev = draws
            if full_layer or last:
                out.append(list(draws.transpose(2, 1, 0)))
        return out if full_layer else out[0]

    def nllik(self, x, y, m=50):
        """Negative predicted log-likelihood of test data under a DGP with ONE likelihood node on top
        (emulation.py:856-914): per imputation the latents' moments at x, the likelihood integrated by
        Gauss-Hermite quadrature (ghdiag), averaged over imputations.  Returns (mean, per-point values)."""
        from .likelihood_class import ghdiag
        if len(self.all_layer[-1]) != 1 or self.all_layer[-1][0].type != 'likelihood':
            raise Exception('The method is only applicable to a DGP with the final layer formed by only ONE node, which '
                            'must be a likelihood node.')
        if self.shard:
            raise NotImplementedError('nllik is evaluated on one rank (emulator(..., shard=False))')
        X0, indices = np.unique(x, return_inverse=True, axis=0)
        indices = np.asarray(indices).reshape(-1)
        if len(X0) != len(x):
            x = X0
        if self.vecch:
            per_layer = self._layer_moments_vecchia(x, m)
        else:
            per_layer = [(a.cpu().numpy(), b.cpu().numpy()) for a, b in self._layer_moments(x)]
        pm, pv = per_layer[-2]
        lik = [ghdiag(self.all_layer[-1][0].pllik, pm[s][indices, :], pv[s][indices, :], y) for s in range(self.N)]
        nl = -np.log(np.mean(lik, axis=0)).flatten()
        return np.mean(nl), nl

    def _predict_points(self, x, method, full_layer, sample_size, m, aggregation):
        """shard='points': this rank predicts its block of rows of x with all N imputations; the blocks are gathered."""
        if method != 'mean_var' or not aggregation:
            raise NotImplementedError("shard='points' covers predict(method='mean_var') with aggregation")
        M = len(x)
        lo, hi = ddist.row_range(M, ddist.rank(), ddist.world())
        xs = x[lo:hi] if hi > lo else x[:1]         # (a rank without rows still takes part in the gather)
        self._in_points = True
        try:
            mu, var = self.predict(xs, method, full_layer, sample_size, m, True)
        finally:
            self._in_points = False
        dev = self.engine.device if ddist.td.get_backend() == 'nccl' else None
        g = lambda a: ddist.allgather_rows(a[:hi - lo], M, dev)
        if full_layer:
            return [g(a) for a in mu], [g(a) for a in var]
        return g(mu), g(var)

    def _predict_vecchia(self, x, full_layer, m, aggregation, method='mean_var', sample_size=50, per_layer=None):
        """Vecchia mode: no stored statistics; every node conditions on its pred_m nearest neighbours
        (kernel_class.py:603-619,647-664) with the imputation's own latents."""
        M, S = len(x), self.N
        if per_layer is None:
            per_layer = self._layer_moments_vecchia(x, m)
        mus, vs = list(per_layer[-1][0]), list(per_layer[-1][1])
        if method == 'sampling':
            return self._draw_samples(per_layer, sample_size, full_layer)
        cat = self._cat()
        if full_layer:
            outm, outv = [], []
            for mu_l, v_l in per_layer:
                mbar = mu_l.mean(0)
                outm.append(mbar)
                outv.append((mu_l ** 2 + v_l).mean(0) - mbar ** 2)
            if cat is not None:
                outm[-1], outv[-1] = cat.prediction(outm[-1], outv[-1])
            return outm, outv
        if not aggregation:
            if cat is not None:
                pr = [cat.prediction(a, b) for a, b in zip(mus, vs)]
                return [p[0] for p in pr], [p[1] for p in pr]
            return mus, vs
        e = self.engine
        s1, s2 = e.zeros(*mus[0].shape), e.zeros(*mus[0].shape)
        for a, b in zip(mus, vs):
            e.moments_accumulate(e.tensor(a), e.tensor(b), s1, s2)
        if self.shard:
            ddist.allreduce_sum(s1, s2)
        e.moments_finalize(self.N_total if self.shard else S, s1, s2)
        if cat is not None:
            return cat.prediction(s1.cpu().numpy(), s2.cpu().numpy())
        return s1.cpu().numpy(), s2.cpu().numpy()

    def ppredict(self, x, method='mean_var', full_layer=False, sample_size=50, m=50, chunk_num=None, core_num=None):
        """emulation.py:578-629 split x over a process pool; test points and imputations already run in parallel on the
        device (`chunk_num` / `core_num` are accepted and unused)."""
        return self.predict(x, method=method, full_layer=full_layer, sample_size=sample_size, m=m)

    @contextlib.contextmanager
    def change_vecch_state(self):
        """Leave-one-out state of the Vecchia prediction branches (emulation.py:90-108): within the context every GP
        node drops the nearest of its conditioning points."""
        gps = [nd for layer in self.all_layer for nd in layer if nd.type == 'gp']
        for nd in gps:
            nd.loo_state = True
        try:
            yield
        finally:
            for nd in gps:
                nd.loo_state = False

    def loo(self, X, method=None, sample_size=50, m=30):
        """Leave-one-out cross validation at the training inputs X (emulation.py:109-143): every GP node conditions on
        its nearest training points with the nearest one (at the first layer the point itself) dropped
        (kernel_class.py:610-611,655-656) -- m of them for a Vecchia emulator, all other n-1 points for a dense one.
        The dense case does not run n factorisations of size n-1 as the reference does: it reuses the emulator's R^-1
        through the rank-one downdate of `_layer_moments_loo`."""
        if method is None:
            method = 'mean_var'
        n_train = len(self.all_layer[0][0].input)
        isrep = len(X) != n_train
        if isrep:
            X, indices = np.unique(X, return_inverse=True, axis=0)
        if not self.vecch:
            per_layer = [(a.cpu().numpy(), b.cpu().numpy()) for a, b in self._layer_moments_loo(X)]
            res = self._predict_vecchia(X, False, m, True, method, sample_size, per_layer=per_layer)
        else:
            with self.change_vecch_state():
                res = self._predict_vecchia(X, False, m + 1, True, method, sample_size)
        if isrep:
            res = type(res)(item[np.asarray(indices).reshape(-1), :] for item in res)
        return res

    def ploo(self, X, method=None, sample_size=50, m=30, core_num=None):
        """emulation.py:146-168 (`core_num` is accepted and unused)."""
        return self.loo(X, method=method, sample_size=sample_size, m=m)

    def metric(self, x_cand, method='ALM', obj=None, nugget_s=1., m=50, score_only=False):
        """Sequential-design criterion at the rows of x_cand (emulation.py:323-420).  ALM (the predictive variance)
        MICE and VIGF (GP hierarchies) are computed from the device layer walk."""
        if x_cand.ndim == 1:
            raise Exception('The candidate design set has to be a numpy 2d-array.')
        lik = any(nd.type == 'likelihood' for nd in self.all_layer[-1])
        L = self.n_layer - 2 if lik else self.n_layer - 1      # the last GP layer carries the criteria (emulation.py:347-420)
        if method == 'ALM':
            if lik:
                sigma2 = self.predict(x=x_cand, full_layer=True, m=m)[1][-2]
            else:
                _, sigma2 = self.predict(x=x_cand, m=m)
            if score_only:
                return sigma2
            idx = np.argmax(sigma2, axis=0)
            return idx, sigma2[idx, np.arange(sigma2.shape[1])]
        if method not in ('MICE', 'VIGF'):
            raise Exception("method must be 'ALM', 'MICE' or 'VIGF'.")
        if self.shard:
            raise NotImplementedError('MICE / VIGF are evaluated on one rank (emulator(..., shard=False))')
        if method == 'VIGF':
            return self._vigf(x_cand, obj, m, score_only)
        # MICE (emulation.py:377-394): mean over imputations of log(predictive variance / smoothed variance of a GP whose
        # design is the candidate set itself, functions.mice_var :244-256)
        if self.vecch:
            per_layer = self._layer_moments_vecchia(x_cand, m)
        else:
            per_layer = [(a.cpu().numpy(), b.cpu().numpy()) for a, b in self._layer_moments(x_cand)]
        sigma2 = per_layer[L][1]
        pred_in = per_layer[L - 1][0] if L > 0 else None
        M, D, S = len(x_cand), len(self.all_layer[L]), self.N
        if lik and self.n_layer == 2:
            # one GP layer under a likelihood (emulation.py:366-375): its predictive variance does not depend on the
            # imputation; the ratio itself is the score
            s_0 = np.stack([self._mice_var(x_cand, x_cand, nd, nugget_s) for nd in self.all_layer[0]], 1)
            avg = sigma2[0] / s_0
        else:
            mice = np.zeros((M, D))
            for i in range(S):
                s_i = np.empty((M, D))
                for k, nd in enumerate(self.all_layer[L]):
                    s_i[:, k] = self._mice_var(x_cand if pred_in is None else pred_in[i], x_cand, nd, nugget_s)
                with np.errstate(divide='ignore'):
                    mice += np.log(sigma2[i] / s_i)
            avg = mice / S
        if score_only:
            return avg
        idx = np.argmax(avg, axis=0)
        return idx, avg[idx, np.arange(avg.shape[1])]

    def _vigf(self, x_cand, obj, m, score_only):
        """VIGF criterion (emulation.py:396-420, predict_vigf :526-576) for GP hierarchies: with b the squared
        difference between the imputation's predictive mean and the output at the nearest training input, and s2 its
        predictive variance, E[b^2 + 6 b s2 + 3 s2^2] - (E[b + s2])^2 over the imputations."""
        if obj is None:
            raise Exception('The dgp object that is used to build the emulator must be supplied to the argument `obj` '
                            'when VIGF criterion is chosen.')
        lik = any(nd.type != 'gp' for nd in self.all_layer[-1])
        if obj.indices is not None and not lik:
            raise Exception('VIGF criterion is currently not applicable to DGP emulators whose training data contain '
                            'replicates but without a likelihood node.')
        L = self.n_layer - 2 if lik else self.n_layer - 1
        X = obj.X
        e = self.engine
        if len(x_cand) * len(X) <= 20_000_000:
            d2 = (x_cand ** 2).sum(1)[:, None] - 2.0 * x_cand @ X.T + (X ** 2).sum(1)[None, :]
            index = np.argmin(d2, axis=1)
        else:
            index = e.nn_query(e.tensor(x_cand), e.tensor(X), 1).cpu().numpy().reshape(-1)
        if self.vecch:
            mean, var = self._layer_moments_vecchia(x_cand, m)[L]
        else:
            mean, var = (t.cpu().numpy() for t in self._layer_moments(x_cand)[L])
        if lik:    # under a likelihood the last GP layer's "outputs" are the imputation's own latents (emulation.py:498-524,567-570)
            Ytr = np.stack([self.latents[s_][L][index, :] for s_ in range(self.N)])                          # (S, M, D)
        else:
            Ytr = np.stack([np.asarray(nd.output, float).reshape(-1)[index] for nd in self.all_layer[-1]], 1)[None]   # (1, M, D)
        bias = (mean - Ytr) ** 2
        E1 = np.mean(bias ** 2 + 6 * bias * var + 3 * var ** 2, axis=0)
        E2 = np.mean(bias + var, axis=0)
        vigf = E1 - E2 ** 2
        if score_only:
            return vigf
        idx = np.argmax(vigf, axis=0)
        return idx, vigf[idx, np.arange(vigf.shape[1])]

    def _mice_var(self, x, x_extra, nd, nugget_s):
        """functions.mice_var (functions.py:244-256) on the device: scale / diag(R^-1) of the correlation matrix of
        the candidate set (smoothing nugget max(nugget_s, nugget)).  pinvh in the reference; a Cholesky-based
        inverse here (R is positive definite for any positive nugget)."""
        e = self.engine
        Xin = x[:, nd.input_dim]
        if nd.connect is not None:
            Xin = np.concatenate((Xin, x_extra[:, nd.connect]), 1)
        n = len(Xin)
        Np = e.padded_dim(n)
        A, Ainv = e.empty(Np, Np), e.empty(Np, Np)
        e.kmatrix(nd.name, e.tensor(Xin), None, None, nd.length, max(nugget_s, nd.nugget[0]), out=A, full=False)
        work = e.potrf_workspace(n, 1)
        _, info = e.potrf(n, A, work=work)
        e.potri(n, A, Ainv, 0, work)
        if int(e.fetch(info)[0]):
            raise np.linalg.LinAlgError('candidate-set correlation matrix is not positive definite')
        d = torch.diagonal(Ainv)[:n]
        return (float(nd.scale[0]) / d).cpu().numpy()

    def pmetric(self, x_cand, method='ALM', obj=None, nugget_s=1., m=50, score_only=False, chunk_num=None, core_num=None):
        """emulation.py:170-321 (`chunk_num` / `core_num` are accepted and unused)."""
        return self.metric(x_cand, method=method, obj=obj, nugget_s=nugget_s, m=m, score_only=score_only)
