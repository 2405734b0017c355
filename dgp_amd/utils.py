"""Small host utilities that keep dgpsi's names (utils.py:51-66, 203-269)."""
import numpy as np
from .kernel_class import bind_private, peek

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


# ---------------------------------------------------------------------------------------------------------------
# Persistence.  dgpsi pickles whole emulator objects with dill (utils.py:18-42).  Here a trained hierarchy is stored
# as plain arrays (.npz, no pickled code), so a structure exported from dgpsi -- save_structure() only touches the
# node attributes both packages share -- loads into this engine and the other way round.
# ---------------------------------------------------------------------------------------------------------------
_NODE_ARRAYS = ('length', 'scale', 'nugget', 'input', 'output', 'global_input', 'input_dim', 'connect', 'prior_coef',
                'bds', 'rep', 'para_path')


def save_structure(all_layer, npz_file):
    """Write the GP hierarchy `all_layer` (e.g. dgp.estimate()) to `npz_file` (arrays only)."""
    out = {'n_layer': np.array(len(all_layer))}
    for l, layer in enumerate(all_layer):
        out['l%d_n' % l] = np.array(len(layer))
        for k, nd in enumerate(layer):
            p = 'l%d_k%d_' % (l, k)
            if getattr(nd, 'type', 'gp') != 'gp':
                if getattr(nd, 'name', None) not in ('Hetero', 'Poisson', 'NegBin', 'ZIP', 'ZINB', 'Categorical'):
                    raise NotImplementedError('unknown likelihood node %r' % getattr(nd, 'name', None))
                out[p + 'likelihood'] = np.array(str(nd.name))
                if nd.name == 'Categorical':
                    out[p + 'cat_num_classes'] = np.array(int(nd.num_classes))
                    out[p + 'cat_link'] = np.array(str(nd.link))
                    out[p + 'cat_eps'] = np.array(float(nd.robustmax_eps))
                    if nd.class_encoder is not None:
                        out[p + 'cat_classes'] = np.asarray(nd.class_encoder.classes_)
                for a in ('input', 'output', 'input_dim', 'rep'):
                    v = getattr(nd, a, None)
                    if v is not None:
                        out[p + a] = np.asarray(v).copy()
                continue
            out[p + 'name'] = np.array(str(nd.name))
            out[p + 'prior_name'] = np.array('' if nd.prior_name is None else str(nd.prior_name))
            out[p + 'flags'] = np.array([bool(nd.scale_est), bool(nd.nugget_est), bool(getattr(nd, 'vecch', False))])
            out[p + 'm'] = np.array(-1 if getattr(nd, 'm', None) is None else int(nd.m))
            for a in _NODE_ARRAYS:
                v = getattr(nd, a, None)
                if v is not None:
                    out[p + a] = np.asarray(v).copy()
    np.savez_compressed(npz_file, **out)


def load_structure(npz_file, engine=None):
    """Rebuild the hierarchy written by save_structure as dgp_amd.kernel nodes (usable by emulator / lgp / dgp)."""
    from .kernel_class import kernel
    with np.load(npz_file if str(npz_file).endswith('.npz') else str(npz_file) + '.npz', allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    layers = []
    for l in range(int(d['n_layer'])):
        layer = []
        for k in range(int(d['l%d_n' % l])):
            p = 'l%d_k%d_' % (l, k)
            g = lambda a: d[p + a].copy() if p + a in d else None   # noqa: E731
            if p + 'likelihood' in d:
                from . import likelihood_class
                if str(d[p + 'likelihood']) == 'Categorical':
                    nd = likelihood_class.Categorical(num_classes=int(d[p + 'cat_num_classes']), input_dim=g('input_dim'),
                                                      link=str(d[p + 'cat_link']), robustmax_eps=float(d[p + 'cat_eps']))
                    if p + 'cat_classes' in d:
                        from sklearn.preprocessing import LabelEncoder
                        nd.class_encoder = LabelEncoder()
                        nd.class_encoder.classes_ = d[p + 'cat_classes'].copy()
                else:
                    nd = getattr(likelihood_class, str(d[p + 'likelihood']))(input_dim=g('input_dim'))
                nd.output, nd.rep = g('output'), g('rep')
                bind_private(nd, 'input', g('input'))
                layer.append(nd)
                continue
            flags = d[p + 'flags']
            prior = str(d[p + 'prior_name']) or None
            nd = kernel(length=g('length'), scale=g('scale'), nugget=g('nugget'), name=str(d[p + 'name']), prior_name=None,
                        bds=g('bds'), nugget_est=bool(flags[1]), scale_est=bool(flags[0]), input_dim=g('input_dim'),
                        connect=g('connect'), engine=engine)
            nd.prior_name, nd.prior_coef = prior, g('prior_coef')   # stored coefficients are the adjusted ones
            if prior == 'ref':
                nd.cl = None
            nd.output = g('output')
            bind_private(nd, 'input', g('input'))
            bind_private(nd, 'global_input', g('global_input'))
            nd.para_path, nd.rep = g('para_path'), g('rep')
            nd.vecch = bool(flags[2])
            nd.m = None if int(d[p + 'm']) < 0 else int(d[p + 'm'])
            nd.D = peek(nd, 'input').shape[1] + (0 if peek(nd, 'global_input') is None else peek(nd, 'global_input').shape[1])
            if nd.rep is not None:
                cnt = np.bincount(nd.rep, minlength=nd.rep.max() + 1)
                nd.W_diag = 1.0 / cnt
            layer.append(nd)
        layers.append(layer)
    return layers


def write(emu, pkl_file):
    """Pickle an emulator / gp / lgp object to `pkl_file`.pkl (dgpsi utils.write, utils.py:18-27).  Device state is
    dropped and rebuilt on first use after read()."""
    import pickle
    with open(pkl_file + '.pkl', 'wb') as f:
        pickle.dump(emu, f)


def read(pkl_file):
    """Load an object stored by write() (dgpsi utils.read, utils.py:30-42)."""
    import pickle
    with open(pkl_file + '.pkl', 'rb') as f:
        return pickle.load(f)


def summary(obj, tablefmt='fancy_grid'):
    """Print the key facts of a kernel / gp / dgp / emulator / lgp object as a table (dgpsi utils.summary,
    utils.py:69-190): per node its type, lengthscales, variance, nugget (with "(fixed)" where not estimated) and wiring."""
    from tabulate import tabulate

    def num(v, est=True):
        txt = np.array2string(np.atleast_1d(v)[0], precision=3, floatmode='fixed')
        return txt if est else txt + ' (fixed)'

    def fun(nd):
        return {'sexp': 'Squared-Exp', 'matern2.5': 'Matern-2.5'}.get(nd.name, nd.name)

    def node_row(nd, head):
        if getattr(nd, 'type', 'gp') == 'likelihood':
            return head + ['Likelihood (%s)' % nd.name, 'NA', 'NA', 'NA',
                           np.array2string(np.asarray(nd.input_dim) + 1, separator=', '), 'NA']
        dims = np.asarray(nd.input_dim) + 1 if nd.input_dim is not None else 'all'
        return head + ['GP (%s)' % fun(nd), np.array2string(nd.length, precision=3, floatmode='fixed', separator=', '),
                       num(nd.scale, nd.scale_est), num(nd.nugget, nd.nugget_est),
                       np.array2string(dims, separator=', ') if not isinstance(dims, str) else dims,
                       'No' if nd.connect is None else np.array2string(np.asarray(nd.connect) + 1, separator=', ')]

    cols = ['Type', 'Length-scale(s)', 'Variance', 'Nugget', 'Input Dims', 'Global Connection']
    kind = type(obj).__name__
    if kind == 'kernel':
        rows = [cols, node_row(obj, [])]
    elif kind == 'gp':
        rows = [cols, node_row(obj.kernel, [])]
    elif kind in ('dgp', 'emulator'):
        if kind == 'dgp' and obj.N != 0:
            print('To get the summary of the trained DGP model, construct an emulator instance using the emulator() class '
                  'and then apply summary() to it.')
            return
        rows = [['Layer No.', 'Node No.'] + cols]
        for l, layer in enumerate(obj.all_layer):
            for k, nd in enumerate(layer):
                rows.append(node_row(nd, ['Layer %d' % (l + 1), 'Node %d' % (k + 1)]))
    elif kind == 'lgp':
        rows = [['Layer No.', 'Emulator No.', 'Type', 'Connection']]
        for l, layer in enumerate(obj.all_layer):
            for k, c in enumerate(layer):
                rows.append(['Layer %d' % (l + 1), 'Emu %d' % (k + 1), c.type.upper(), str(c.local_input_idx)])
    else:
        raise Exception('summary() takes a kernel, gp, dgp, emulator or lgp object.')
    print(tabulate(rows, headers='firstrow', tablefmt=tablefmt))
