"""Sample paths of a deep GP prior -- mirror of dgpsi.synthetic.path (synthetic.py:6-60): layer by layer every node
draws f = chol(scale (K + nugget I)) z at the outputs of the layer before, z from numpy's global stream like the
reference.  Assembly, factorisation and the triangular product run on the device."""
import copy

import numpy as np
from .kernel_class import bind_private, peek

from .ops import default_engine


class path:
    def __init__(self, X, all_layer, device=None):
        self.X = X
        self.n_layer = len(all_layer)
        self.all_layer = copy.deepcopy(all_layer)
        self.engine = default_engine(device)
        for layer in self.all_layer:
            for nd in layer:
                if nd.connect is not None:
                    bind_private(nd, 'global_input', self.X[:, nd.connect].copy())

    @staticmethod
    def k_matrix(X, length, name):
        """Correlation matrix with a unit diagonal as a numpy array (synthetic.py:46-60); `generate` assembles its
        matrices itself and does not call this."""
        e = default_engine(None)
        n = len(X)
        A = e.kmatrix(name, e.tensor(np.asarray(X, float)), None, None, np.atleast_1d(np.asarray(length, float)), 0.0, full=True)
        return A[:n, :n].cpu().numpy()

    def generate(self, N):
        """N sample paths at the rows of X: array (D_out, N, n)."""
        e = self.engine
        n = len(self.X)
        out_dim = len(self.all_layer[-1])
        rec = np.empty((N, n, out_dim))
        Np = e.padded_dim(n)
        A = e.empty(Np, Np)
        work = e.potrf_workspace(n, 1)
        for i in range(N):
            x = self.X
            for layer in self.all_layer:
                out = np.empty((n, len(layer)))
                for k, nd in enumerate(layer):
                    In = x if nd.input_dim is None else x[:, nd.input_dim]
                    if nd.connect is not None:
                        In = np.concatenate((In, peek(nd, 'global_input')), 1)
                    e.kmatrix(nd.name, e.tensor(In), None, None, nd.length, nd.nugget[0], out=A, full=False)
                    _, info = e.potrf(n, A, work=work)
                    if int(e.fetch(info)[0]):
                        raise np.linalg.LinAlgError('Matrix is not positive definite')
                    z = np.random.normal(size=[n, 1])
                    out[:, k] = e.trmv_lower(n, A, [float(nd.scale[0])], e.tensor(z.reshape(1, n))).cpu().numpy()[0]
                x = out
            rec[i] = x
        return rec.transpose(2, 0, 1)
