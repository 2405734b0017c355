"""Lock-step evaluation of the M-step objectives.

Every GP node runs its own scipy L-BFGS-B (one Python thread each, as the reference runs them one after another,
dgp.py:1391-1398); each objective evaluation needs K assembly + Cholesky + inverse + reductions on the device.
Measured on MI355X, independent factorisations on separate HIP streams overlap only ~3x: the ~100 launches of one
evaluation are serialised on the host side of the runtime.  So the threads rendezvous instead: when every still
active optimiser has asked for an evaluation, the last one to arrive runs ONE batched pipeline for all of them
(per-node K assembly with its own hyper-parameters, batched potrf / potri graphs, per-node reductions, a single
device-to-host copy) and hands the results back.  Results are those of kernel.llik evaluated alone.
"""
import threading

import numpy as np
import torch


class LlikBatcher:
    def __init__(self, engine, n_active):
        self.e = engine
        self.cv = threading.Condition()
        self.pending = []
        self.results = {}
        self.active = int(n_active)
        self.error = None
        self.rounds = 0
        self.evals = 0

    # ---- called from the optimiser threads -------------------------------------------------
    def evaluate(self, node):
        with self.cv:
            self.pending.append(node)
            if len(self.pending) >= self.active:
                self._run()
            else:
                while id(node) not in self.results and self.error is None:
                    self.cv.wait()
            if self.error is not None:
                raise self.error
            return self.results.pop(id(node))

    def done(self):
        """An optimiser finished (or failed): it no longer takes part in the rendezvous."""
        with self.cv:
            self.active -= 1
            if self.pending and len(self.pending) >= self.active:
                self._run()

    # ---- the batched pipeline (lock held by the thread that completes the rendezvous) --------
    def _run(self):
        try:
            groups = {}
            for nd in self.pending:
                groups.setdefault(len(nd.output), []).append(nd)
            for n, nodes in groups.items():
                self._run_group(n, nodes)
            self.rounds += 1
            self.evals += len(self.pending)
        except Exception as ex:   # surfaces in every waiting optimiser
            self.error = ex
        self.pending = []
        self.cv.notify_all()

    def _run_group(self, n, nodes):
        e = self.e
        B = len(nodes)
        Np = e.padded_dim(n)
        with e.stream():
            cap = max(B, getattr(self, '_cap', 0))
            self._cap = cap
            A = e.workspace(('mstepA', n), cap * Np * Np * 8).view(torch.float64)[:B * Np * Np].view(B, Np, Np)
            Ainv = e.workspace(('mstepAinv', n), cap * Np * Np * 8).view(torch.float64)[:B * Np * Np].view(B, Np, Np)
            for b, nd in enumerate(nodes):
                s = nd._staged if nd._staged is not None else nd._stage()
                e.kmatrix(nd.name, s['Xl'], None, s['Xg'], nd.length, nd.nugget[0], W=s['W'], out=A[b], full=False, Y=s['y'])
            work = e.potrf_workspace(n, B)
            logdet, info = e.potrf(n, A, batch=B, work=work)
            quad = e.aug_quad(n, A, B, 1)
            e.potri(n, A, Ainv, 1, work, batch=B)
            reds = []
            for b, nd in enumerate(nodes):
                s = nd._staged
                red, P = e.grad_reduce(nd.name, s['Xl'], None, s['Xg'], nd.length, nd.nugget[0], nd.nugget_est, Ainv[b], W=s['W'])
                reds.append((red, P))
            packed = torch.cat([logdet, quad.reshape(-1), info.to(torch.float64)] + [r for r, _ in reds]).cpu().numpy()
        off = 3 * B
        for b, (nd, (_, P)) in enumerate(zip(nodes, reds)):
            red = packed[off:off + 2 * P]
            off += 2 * P
            self.results[id(nd)] = np.concatenate(([packed[b], packed[B + b]], red, [packed[2 * B + b]]))
