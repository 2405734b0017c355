"""Linked (D)GP emulation of a feed-forward system of emulators -- mirror of dgpsi.container / dgpsi.lgp
(linkgp.py:12-608), mean/variance prediction.  Pure orchestration over kernel.gp_prediction /
linkgp_prediction / linkgp_prediction_full; aggregation over imputations as emulation.py:846-847."""
import contextlib
import copy

import numpy as np

from .imputation import imputer


class container:
    """A trained GP (`gp.export()`) or DGP (`dgp.estimate()`) with its wiring into the system
    (linkgp.py:12-56).  local_input_idx: indices of the feeding layer's outputs (1d-array), or a list with one
    entry (array or None) per preceding layer."""

    def __init__(self, structure, local_input_idx=None, block=True):
        if len(structure) == 1:
            self.type, self.structure = 'gp', structure[0]
            self.vecch = bool(self.structure.vecch)
        else:
            self.type, self.structure = 'dgp', structure
            self.vecch = bool(structure[0][0].vecch)
            self.imp = imputer(self.structure, block)
            if self.vecch:
                self.imp.update_ord_nn()
            self.imp.sample(burnin=50)
        self.local_input_idx = local_input_idx

    def _gp_nodes(self):
        return [self.structure] if self.type == 'gp' else [nd for layer in self.structure for nd in layer if nd.type == 'gp']

    def to_vecchia(self):
        """Vecchia predictions for this emulator (linkgp.py:64-75)."""
        if not self.vecch:
            self.vecch = True
            for nd in self._gp_nodes():
                nd.vecch = True

    def remove_vecchia(self):
        """Dense predictions (linkgp.py:77-89); the n x n statistics are rebuilt on first use."""
        if self.vecch:
            self.vecch = False
            for nd in self._gp_nodes():
                nd.vecch = False
                nd._stats = None

    def set_local_input(self, idx, new=False):
        if not new:
            self.local_input_idx = idx
            return None
        c = copy.copy(self)
        c.local_input_idx = idx
        return c

    def __copy__(self):
        c = type(self).__new__(type(self))
        c.__dict__.update(self.__dict__)
        c.local_input_idx = copy.copy(self.local_input_idx)
        return c

    def _snapshot(self):
        """Copy holding the current imputation (arrays only; device statistics are rebuilt on first use)."""
        c = type(self).__new__(type(self))
        c.type, c.vecch, c.local_input_idx = self.type, self.vecch, copy.copy(self.local_input_idx)
        c.structure = copy.deepcopy(self.structure)
        nodes = [c.structure] if c.type == 'gp' else [nd for layer in c.structure for nd in layer]
        src = [self.structure] if self.type == 'gp' else [nd for layer in self.structure for nd in layer]
        for a, b in zip(nodes, src):
            if getattr(b, 'type', None) == 'gp':
                a.engine = b.engine
        return c


def _ensure_stats(nd):
    if nd.type == 'gp' and not nd.vecch and nd._stats is None:
        nd.compute_stats()


class lgp:
    """all_layer: list of layers of containers; N imputations (1 if the system has GP emulators only)  (linkgp.py:140-165)."""

    def __init__(self, all_layer, N=10):
        self.L = len(all_layer)
        self.all_layer = all_layer
        self.num_model = [len(layer) for layer in all_layer[1:]]
        if not any(c.type == 'dgp' for layer in all_layer for c in layer):
            N = 1
        self.all_layer_set = []
        for _ in range(N):
            one = []
            for layer in all_layer:
                row = []
                for c in layer:
                    if c.type == 'dgp':
                        if c.vecch:
                            c.imp.update_ord_nn()
                        c.imp.sample()
                    row.append(c._snapshot())
                one.append(row)
            self.all_layer_set.append(one)

    def set_vecchia(self, mode):
        """Vecchia (True) or dense (False) predictions for all emulators of the system, or per emulator with a list
        shaped like all_layer (linkgp.py:180-212)."""
        if isinstance(mode, list):
            if len(mode) != len(self.all_layer) or any(len(a) != len(b) for a, b in zip(mode, self.all_layer)):
                raise Exception('mode has a different shape as all_layer.')
        else:
            mode = [[mode for _ in layer] for layer in self.all_layer]
        for system in [self.all_layer] + list(self.all_layer_set):
            for layer, ml in zip(system, mode):
                for c, on in zip(layer, ml):
                    c.to_vecchia() if on else c.remove_vecchia()

    # -------------------------------------------------------------- single emulators
    @staticmethod
    def gp_pred(x, m, v, z, structure, m_pred):
        """GP emulator with deterministic (x) or Gaussian (m, v) inputs (linkgp.py:503-515)."""
        structure.pred_m = m_pred
        _ensure_stats(structure)
        if x is None:
            mu, s2 = structure.linkgp_prediction(m=m, v=v, z=z)
        else:
            mu, s2 = structure.gp_prediction(x=x, z=z)
        return mu.reshape(-1, 1), s2.reshape(-1, 1)

    @staticmethod
    def dgp_pred(x, m, v, z, structure, pred_m):
        """Layer walk through a DGP emulator whose input is deterministic (x) or Gaussian (m, v [+ external z])
        (linkgp.py:517-608, GP nodes).  Returns (mean, var) of the layer before last and of the last layer."""
        M = len(m) if x is None else len(x)
        L = len(structure)
        internal, external = structure[0][0].input_dim, structure[0][0].connect
        mean_in = var_in = None
        for l, layer in enumerate(structure):
            mo, vo = np.empty((M, len(layer))), np.empty((M, len(layer)))
            for k, nd in enumerate(layer):
                if nd.type != 'gp':   # likelihood node on top of the emulator (linkgp.py:574-576)
                    mo[:, k], vo[:, k] = nd.prediction(m=mean_in[:, nd.input_dim], v=var_in[:, nd.input_dim])
                    continue
                nd.pred_m = pred_m
                _ensure_stats(nd)
                if l == 0:
                    mo[:, k], vo[:, k] = nd.linkgp_prediction(m=m, v=v, z=z) if x is None else nd.gp_prediction(x=x, z=z)
                    continue
                mk, vk = mean_in[:, nd.input_dim], var_in[:, nd.input_dim]
                if nd.connect is None:
                    mo[:, k], vo[:, k] = nd.linkgp_prediction(m=mk, v=vk, z=None)
                elif x is not None:
                    mo[:, k], vo[:, k] = nd.linkgp_prediction(m=mk, v=vk, z=x[:, nd.connect])
                else:
                    # the node's global inputs are themselves uncertain (outputs of feeding emulators) and/or external
                    if l == L - 1:
                        i1 = np.where(nd.connect[:, None] == internal[None, :])[1]
                        i2 = np.array([], dtype=int) if external is None else np.where(nd.connect[:, None] == external[None, :])[1]
                    else:
                        D = m.shape[1]
                        i1, i2 = nd.connect[nd.connect <= D - 1], nd.connect[nd.connect > D - 1] - D
                    if i1.size == 0:
                        mo[:, k], vo[:, k] = nd.linkgp_prediction(m=mk, v=vk, z=z[:, i2])
                    else:
                        mo[:, k], vo[:, k] = nd.linkgp_prediction_full(m=mk, v=vk, m_z=m[:, i1], v_z=v[:, i1],
                                                                        z=None if i2.size == 0 else z[:, i2])
            if l < L - 1:
                mean_in, var_in = mo, vo
        return mean_in, var_in, mo, vo

    def _emulate(self, model, x, m, v, z, pred_m, before=False):
        if model.type == 'gp':
            out = self.gp_pred(x, m, v, z, model.structure, pred_m)
            return (None, None) + out if before else out
        out = self.dgp_pred(x, m, v, z, model.structure, pred_m)
        return out if before else out[2:]

    @staticmethod
    def _draw(model, mk, vk, m_before, v_before, sample_size):
        """sample_size draws per test point from one emulator's predictive distributions, (q, M, sample_size)
        (linkgp.py:383-386,408-421).  A likelihood node on top of a DGP emulator samples y from draws of its feeding
        latents.  (For the GP nodes of a DGP emulator in the last layer the reference takes the spread from the layer
        before, linkgp.py:416 -- a slip that fails as soon as the two layers differ in width; the nodes' own
        predictive variances are used here.)"""
        M, q = mk.shape
        if model.type == 'gp' or all(nd.type == 'gp' for nd in model.structure[-1]):
            return np.random.normal(mk, np.sqrt(vk), size=(sample_size, M, q)).transpose(2, 1, 0)
        out = np.empty((q, M, sample_size))
        for c, nd in enumerate(model.structure[-1]):
            if nd.type == 'gp':
                out[c] = np.random.normal(mk[:, [c]], np.sqrt(vk[:, [c]]), size=(M, sample_size))
            else:
                lat = np.random.normal(m_before, np.sqrt(v_before), size=(sample_size,) + m_before.shape)
                out[c] = np.array([nd.sampling(lat[i][:, nd.input_dim]) for i in range(sample_size)]).T
        return out

    # -------------------------------------------------------------- the system
    def predict(self, x, method='mean_var', full_layer=False, sample_size=50, m=50):
        """Means and variances of the final-layer emulators' outputs (lists of (M x q) arrays), or of every
        layer if full_layer (linkgp.py:285-501); method='sampling': per emulator an array (q, M, N * sample_size) of
        draws from the imputations' predictive distributions."""
        if method not in ('mean_var', 'sampling'):
            raise Exception("method must be either 'mean_var' or 'sampling'.")
        sampling = method == 'sampling'
        if isinstance(x, list):
            if len(x) != self.L:
                raise Exception('When test input is given as a list, it must contain global inputs to the all layers '
                                '(even with no global inputs to internal layers). Set None as the global input to the '
                                'internal models if they have no global inputs.')
        else:
            if x.ndim == 1:
                raise Exception('The testing input has to be a numpy 2d-array.')
            x = [x] + [[None] * k for k in self.num_model]
        means, variances, draws = [], [], []
        for one in self.all_layer_set:
            feed_m, feed_v, lay_m, lay_v, lay_s = [], [], [], [], []
            for l, layer in enumerate(one):
                ms, vs, ss = [], [], []
                for k, model in enumerate(layer):
                    if l == 0:
                        if isinstance(model.local_input_idx, list):
                            raise Exception('When an emulator is in the first layer, local_input_idx must be a 1d-array.')
                        mb, vb, mk, vk = self._emulate(model, x[0][:, model.local_input_idx], None, None, None, m, before=True)
                    else:
                        idx = model.local_input_idx
                        if not isinstance(idx, list):
                            idx = [None] * (l - 1) + [idx]
                        elif len(idx) != l:
                            raise Exception('local_input_idx should be a list that has length of %i.' % l)
                        m_in = np.concatenate([feed_m[i][:, j] for i, j in enumerate(idx) if j is not None], axis=1)
                        v_in = np.concatenate([feed_v[i][:, j] for i, j in enumerate(idx) if j is not None], axis=1)
                        mb, vb, mk, vk = self._emulate(model, None, m_in, v_in, x[l][k], m, before=True)
                    ms.append(mk)
                    vs.append(vk)
                    if sampling and (full_layer or l == self.L - 1):
                        ss.append(self._draw(model, mk, vk, mb, vb, sample_size))
                lay_m.append(ms)
                lay_v.append(vs)
                lay_s.append(ss)
                feed_m.append(np.concatenate(ms, axis=1))
                feed_v.append(np.concatenate(vs, axis=1))
            means.append(lay_m if full_layer else lay_m[-1])
            variances.append(lay_v if full_layer else lay_v[-1])
            draws.append(lay_s if full_layer else lay_s[-1])
        if sampling:    # per emulator (q, M, S * sample_size): the imputations' draws side by side (linkgp.py:496-500)
            if full_layer:
                return [[np.concatenate([draws[s_][l][k] for s_ in range(len(draws))], axis=2)
                         for k in range(len(self.all_layer[l]))] for l in range(self.L)]
            return [np.concatenate([draws[s_][k] for s_ in range(len(draws))], axis=2) for k in range(len(self.all_layer[-1]))]

        def agg(ms, vs):   # emulation.py:846-847 over the imputations
            ms, vs = np.asarray(ms), np.asarray(vs)
            mu = ms.mean(0)
            return mu, (ms ** 2 + vs).mean(0) - mu ** 2
        if full_layer:
            out = [[agg([means[s][l][k] for s in range(len(means))], [variances[s][l][k] for s in range(len(means))])
                    for k in range(len(self.all_layer[l]))] for l in range(self.L)]
            return [[o[0] for o in row] for row in out], [[o[1] for o in row] for row in out]
        out = [agg([means[s][k] for s in range(len(means))], [variances[s][k] for s in range(len(means))])
               for k in range(len(self.all_layer[-1]))]
        return [o[0] for o in out], [o[1] for o in out]

    @contextlib.contextmanager
    def temp_all_layer(self):
        """A deep copy of the linked structure to work on (linkgp.py:172-178)."""
        yield copy.deepcopy(self.all_layer)

    def ppredict(self, x, method='mean_var', full_layer=False, sample_size=50, m=50, chunk_num=None, core_num=None):
        """linkgp.py:214-262 (`chunk_num` / `core_num` are accepted and unused)."""
        return self.predict(x, method=method, full_layer=full_layer, sample_size=sample_size, m=m)
