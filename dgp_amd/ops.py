"""Typed Python front of the C-ABI (include/dgp_amd.h).

`Engine` owns one libdgp_amd context bound to a device and a HIP stream.  Device
memory is plain torch tensors (torch is plumbing here: allocation, streams,
torch.distributed); every numerical operation is a HIP kernel behind the C-ABI.
Host-side small parameters (lengthscales, angles) are numpy arrays.
"""
import ctypes as C

import numpy as np
import torch

import os

from . import _lib
from ._lib import lib, KIND

_NO_PINNED_UPLOAD = os.environ.get('DGPAMD_PINNED_UPLOAD') == '0'   # (comparison runs: pageable, blocking uploads)
_POISON_ALL = os.environ.get('DGPAMD_POISON_LDS') == '2'
_POISON_HBM = os.environ.get('DGPAMD_POISON_HBM') == '1'   # (debugging: fresh device buffers are filled with 0xFF bytes -- NaNs as doubles)   # (debugging: NaNs into every CU's LDS before every library call)


class DgpAmdError(RuntimeError):
    pass


class HandoffError(DgpAmdError):
    """The one-launch factorisation gave up waiting for another workgroup (its workgroups were not co-resident: a device
    shared with another process, a debugger, a time-sliced partition).  Not a numerical failure; the per-block-step
    factorisation (Engine.set_potrf_mode(0)) has no such waits -- dgp.train retries the iteration once through it."""


def raise_not_pd(info):
    """A non-zero factorisation status as the exception it stands for.  info > 0: LAPACK's index of the first
    non-positive pivot -> numpy.linalg.LinAlgError, the reference's signal (dgp.train restarts on it, compute_stats
    falls back to pinvh).  info < 0: a bounded in-kernel spin gave up (a lost workgroup hand-off) -- a device problem,
    never a numerical one, so it must not be swallowed by those recovery paths: DgpAmdError."""
    info = int(info)
    if info < 0:
        raise HandoffError('factorisation: an in-kernel hand-off timed out (info = %d); this is not a numerical failure' % info)
    raise np.linalg.LinAlgError('%d-th leading minor of the array is not positive definite' % info)


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))


class Engine:
    def __init__(self, device=0, stream=None):
        if not torch.cuda.is_available():
            raise DgpAmdError('dgp_amd needs a HIP device: torch.cuda.is_available() is False and there is no CPU path')
        self.device = torch.device('cuda', device)
        torch.cuda.set_device(self.device)
        if stream is None:
            # Use the thread's CURRENT torch stream so that torch's copies / allocator and the library's launches
            # are ordered on one stream (every engine created in a thread without an explicit stream shares it).
            # hipGraph capture is illegal on the null stream: if that is the current one, switch the thread to a
            # side stream first.
            stream = torch.cuda.current_stream(self.device)
            if stream.cuda_stream == 0:
                stream = torch.cuda.Stream(self.device)
                torch.cuda.set_stream(stream)
        self._torch_stream = stream
        h = C.c_void_p()
        rc = lib.dgpamd_create(int(device), C.c_void_p(self._torch_stream.cuda_stream), C.byref(h))
        if rc != 0:
            raise DgpAmdError('dgpamd_create failed (rc=%d)' % rc)
        self.h = h
        self._ws = {}

    def close(self):
        if getattr(self, 'h', None):
            lib.dgpamd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ utils
    def _enter(self):
        """Called before every library call.  The library launches on THIS engine's stream; torch work the caller
        queued on another current stream (a different thread, a user's torch.cuda.stream block) is ordered before it by
        an event wait, and _chk() orders the caller's stream after the call again.  The creating thread's current
        stream IS the engine's stream, so both are no-ops there.  Returns None (used as `_enter() or lib.f(...)`)."""
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != self._torch_stream.cuda_stream:
            self._torch_stream.wait_stream(cur)
            self._foreign = cur
        if _POISON_ALL:
            lib.dgpamd_debug_poison_lds(self.h)
        return None

    def _chk(self, rc):
        f = getattr(self, '_foreign', None)
        if f is not None:
            self._foreign = None
            f.wait_stream(self._torch_stream)
        if rc != 0:
            exc = self.__dict__.pop('_reduce_exc', None)
            if exc is not None:   # (an exception inside the reduce hook: the library call it interrupted has failed with it)
                raise exc
            raise DgpAmdError('libdgp_amd rc=%d: %s' % (rc, lib.dgpamd_last_error(self.h).decode()))

    def sync(self):
        self._chk(lib.dgpamd_sync(self.h))

    def set_linkgp_direct(self, enable):
        """Matern linked-GP J factor: reference's direct expression (True) or its separable form (default)."""
        self._chk(self._enter() or lib.dgpamd_set_linkgp_direct(self.h, 1 if enable else 0))

    def set_graphs(self, enable):
        self._chk(self._enter() or lib.dgpamd_set_graphs(self.h, 1 if enable else 0))

    def set_potrf_mode(self, mode):
        """1 (default): the factorisation is one persistent dataflow launch; 0: one launch per 64-column block step."""
        self._chk(self._enter() or lib.dgpamd_set_potrf_mode(self.h, int(mode)))

    def stream(self):
        """Context manager making this engine's HIP stream torch's current stream (so that torch's
        allocator and copies are ordered with the library's launches)."""
        return torch.cuda.stream(self._torch_stream)

    def tensor(self, a, dtype=torch.float64):
        """numpy -> device.  Through page-locked memory of torch's caching host allocator and an asynchronous copy: the call
        returns when the bytes are staged, not when the stream has reached the copy (a pageable upload waits for every kernel
        queued before it -- 1 to 2.5 ms per call inside the Vecchia I-step's set-up at n = 50 000)."""
        a = np.asarray(a)
        if self.device.type != 'cuda' or a.size == 0 or _NO_PINNED_UPLOAD:
            a = np.ascontiguousarray(a)
            if not a.flags.writeable:
                a = a.copy()
            return torch.as_tensor(a, dtype=dtype).to(self.device, non_blocking=False)
        pin = torch.empty(a.shape, dtype=dtype, pin_memory=True)
        np.copyto(pin.numpy(), a, casting='unsafe')
        return pin.to(self.device, non_blocking=True)   # (the allocator keeps the block until the copy has run)

    def empty(self, *shape, dtype=torch.float64):
        t = torch.empty(*shape, dtype=dtype, device=self.device)
        if _POISON_HBM and t.numel():   # (debugging: whatever is read before it is written shows up as NaN / -1)
            t.view(torch.uint8).fill_(255)
        return t

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def workspace(self, key, nbytes):
        """Cached byte workspace (grown on demand) -- no allocation inside hot loops."""
        t = self._ws.get(key)
        if t is None or t.numel() < nbytes:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            if _POISON_HBM:
                t.fill_(255)
            self._ws[key] = t
        return t

    @staticmethod
    def padded_dim(n):
        return int(lib.dgpamd_padded_dim(int(n)))

    def event(self):
        e = C.c_void_p()
        self._chk(self._enter() or lib.dgpamd_event_create(self.h, C.byref(e)))
        return e

    def record(self, ev):
        self._chk(self._enter() or lib.dgpamd_event_record(self.h, ev))

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        self._chk(self._enter() or lib.dgpamd_event_elapsed_ms(self.h, start, stop, C.byref(ms)))
        return float(ms.value)

    PROF = dict(kmatrix=1, potrf_diag=2, trsm=3, syrk=4, trtri=5, lauum=6, grad=7, linkgp_j=8, gp_quad=9)

    def prof_enable(self, name):
        self._chk(self._enter() or lib.dgpamd_prof_enable(self.h, self.PROF[name] if name else 0))

    def prof_event_overhead_us(self):
        v = C.c_double()
        self._chk(self._enter() or lib.dgpamd_prof_event_overhead_us(self.h, C.byref(v)))
        return float(v.value)

    def prof_collect(self):
        """(launches, total_ms, algorithmic work) of the launches timed since prof_enable."""
        n, ms, w = C.c_int64(), C.c_double(), C.c_double()
        self._chk(self._enter() or lib.dgpamd_prof_collect(self.h, C.byref(n), C.byref(ms), C.byref(w)))
        return int(n.value), float(ms.value), float(w.value)

    # --------------------------------------------------------------- kernels
    @staticmethod
    def _colmap(colmap, Dl):
        if colmap is None:
            return None, None
        cm = np.ascontiguousarray(np.asarray(colmap, dtype=np.int32))
        assert cm.shape == (Dl,)
        return cm, _hp(cm)

    def kmatrix(self, kind, Xloc, colmap, Xglob, length, nugget, W=None, out=None, full=True, Y=None, batch=1):
        """K (full=True: n x n, both triangles; full=False: augmented Np x Np buffer with rows of Y).
        Xloc: (batch?, n, ldloc) tensor; colmap: gathered columns (None = all)."""
        n, ldloc = Xloc.shape[-2], Xloc.shape[-1]
        Dl = ldloc if colmap is None else len(colmap)
        Dg = 0 if Xglob is None else Xglob.shape[1]
        cm, cmp_ = self._colmap(colmap, Dl)
        length = _f64(length)
        stride_loc = n * ldloc if Xloc.dim() == 3 else 0
        if full:
            # (a caller's `out` may be a view of wider rows, e.g. buf[:, :n] of an (n, ld) buffer with 8 ld a multiple of 128 bytes:
            #  rows that start on 128-byte lines let every tile store whole lines -- 0.55 -> 0.72 of HBM for the stores alone at
            #  n = 5000, profiles/r05_kmatrix_store_shapes.txt)
            ld = n if out is None else int(out.stride(-2))
            r = 0
            if out is not None:   # (a transposed or narrow view would be written in a layout the caller does not expect)
                if out.stride(-1) != 1 or ld < n or out.shape[-1] < n or out.shape[-2] < n:
                    raise ValueError('kmatrix: `out` needs unit column stride and rows of at least n entries (strides %s, n = %d)' % (tuple(out.stride()), n))
                if batch > 1 and (out.dim() != 3 or out.stride(0) < n * ld):
                    raise ValueError('kmatrix: a batched `out` is (batch, n, ld) with matrices that do not overlap')
        else:
            ld = self.padded_dim(n)
            r = 0 if Y is None else (1 if Y.dim() == 1 else Y.shape[-2])
        if out is None:
            out = self.empty((batch, ld, ld) if batch > 1 else (ld, ld))
        stride_k = (ld * ld if (out.dim() < 3 or not full) else int(out.stride(0))) if batch > 1 else 0
        stride_y = 0
        if Y is not None and Y.dim() == 3:
            stride_y = Y.shape[-2] * Y.shape[-1]
        self._chk(self._enter() or lib.dgpamd_kmatrix(self.h, KIND[kind], n, _dp(Xloc), ldloc, stride_loc, cmp_, Dl, _dp(Xglob), Dg,
                                     _hp(length), len(length), float(nugget), _dp(W), _dp(out), ld, stride_k,
                                     1 if full else 0, _dp(Y), n, stride_y, r, batch))
        return out

    def potrf_workspace(self, n, batch):
        return self.workspace(('potrf', n, batch), lib.dgpamd_potrf_workspace(n, batch))

    def potrf(self, n, A, batch=1, work=None):
        """In-place factorisation of augmented buffers.  Returns (logdet, info) device tensors."""
        Np = self.padded_dim(n)
        if work is None:
            work = self.potrf_workspace(n, batch)
        logdet = self.empty(batch)
        info = self.empty(batch, dtype=torch.int32)
        self._chk(self._enter() or lib.dgpamd_potrf(self.h, n, _dp(A), Np * Np, batch, _dp(logdet), _dp(info), _dp(work)))
        return logdet, info

    def potrf_inv(self, n, A, T, S, batch=1, work=None):
        """Factorisation and inverse in one sweep (dgpamd_potrf_inv): A -> L, T -> L^-T, S -> K^-1 (lower tiles,
        row n = -alpha^T).  Returns (logdet, info) device tensors."""
        Np = self.padded_dim(n)
        if work is None:
            work = self.potrf_workspace(n, batch)
        logdet = self.empty(batch)
        info = self.empty(batch, dtype=torch.int32)
        self._chk(self._enter() or lib.dgpamd_potrf_inv(self.h, n, _dp(A), _dp(T), _dp(S), Np * Np, batch, _dp(logdet), _dp(info), _dp(work)))
        return logdet, info

    def aug_quad(self, n, A, batch, r):
        Np = self.padded_dim(n)
        out = self.empty(batch, r, r)
        self._chk(self._enter() or lib.dgpamd_aug_quad(self.h, n, _dp(A), Np * Np, batch, r, _dp(out)))
        return out

    def loglik_finish(self, n, A, logdet, scale, batch=1):
        """ll (batch,) of buffers assembled with their y row and factored by another call (dgpamd_loglik_finish)."""
        Np = self.padded_dim(n)
        ll = self.empty(batch)
        self._chk(self._enter() or lib.dgpamd_loglik_finish(self.h, n, _dp(A), Np * Np, batch, _dp(logdet), float(scale), _dp(ll)))
        return ll

    def loglik(self, kind, Xloc, colmap, Xglob, length, nugget, scale, y, W=None, batch=1, A=None, ll=None, info=None):
        """Batched ESS target (kernel_class.py:481-492).  Returns (ll, info) device tensors (no sync)."""
        n, ldloc = Xloc.shape[-2], Xloc.shape[-1]
        Dl = ldloc if colmap is None else len(colmap)
        Dg = 0 if Xglob is None else Xglob.shape[1]
        cm, cmp_ = self._colmap(colmap, Dl)
        length = _f64(length)
        Np = self.padded_dim(n)
        if A is None:
            A = self.workspace(('loglikA', n, batch), batch * Np * Np * 8)
        work = self.potrf_workspace(n, batch)
        if ll is None:
            ll = self.empty(batch)
        if info is None:
            info = self.empty(batch, dtype=torch.int32)
        stride_loc = n * ldloc if Xloc.dim() == 3 else 0
        self._chk(self._enter() or lib.dgpamd_loglik(self.h, KIND[kind], n, _dp(Xloc), ldloc, stride_loc, cmp_, Dl, _dp(Xglob), Dg,
                                    _hp(length), len(length), float(nugget), _dp(W), float(scale), _dp(y), _dp(A),
                                    Np * Np, batch, _dp(ll), _dp(info), _dp(work)))
        return ll, info

    def trmv_lower(self, n, L, scale, z, batch=1, out=None):
        Np = self.padded_dim(n)
        sc = _f64(scale)
        if len(sc) == 1 and batch > 1:
            sc = np.repeat(sc, batch)
        if out is None:
            out = self.empty(batch, n)
        self._chk(self._enter() or lib.dgpamd_trmv_lower(self.h, n, _dp(L), Np * Np, _hp(sc), _dp(z), _dp(out), batch))
        return out

    def ess_propose(self, F, NU, thetas, out=None):
        n, M = F.shape
        th = _f64(thetas)
        B = len(th)
        if out is None:
            out = self.empty(B, n, M)
        self._chk(self._enter() or lib.dgpamd_ess_propose(self.h, n, M, _dp(F), _dp(NU), _hp(th), B, _dp(out)))
        return out

    def potri(self, n, A, Ainv, r, work, batch=1):
        Np = self.padded_dim(n)
        self._chk(self._enter() or lib.dgpamd_potri_batched(self.h, n, _dp(A), _dp(Ainv), Np * Np if batch > 1 else 0, r, batch, _dp(work)))
        return Ainv

    def grad_reduce(self, kind, Xloc, colmap, Xglob, length, nugget, nugget_est, Ainv, W=None):
        n, ldloc = Xloc.shape[-2], Xloc.shape[-1]
        Dl = ldloc if colmap is None else len(colmap)
        Dg = 0 if Xglob is None else Xglob.shape[1]
        cm, cmp_ = self._colmap(colmap, Dl)
        length = _f64(length)
        P = (1 if len(length) == 1 else Dl + Dg) + (1 if nugget_est else 0)
        work = self.workspace(('grad', n, P), lib.dgpamd_grad_workspace(n, P))
        out = self.empty(2 * P)
        self._chk(self._enter() or lib.dgpamd_grad_reduce(self.h, KIND[kind], n, _dp(Xloc), ldloc, cmp_, Dl, _dp(Xglob), Dg, _hp(length),
                                         len(length), float(nugget), _dp(W), 1 if nugget_est else 0, _dp(Ainv), _dp(out),
                                         _dp(work)))
        return out, P

    def gemv(self, A, x, out=None):
        rows, cols = A.shape
        if out is None:
            out = self.empty(rows)
        self._chk(self._enter() or lib.dgpamd_gemv(self.h, rows, cols, _dp(A), A.stride(0), _dp(x), _dp(out)))
        return out

    def pinvh(self, K):
        """Pseudo-inverse of the symmetric matrix K (n x n device tensor) with scipy.linalg.pinvh's rule: eigenvalues
        of magnitude <= n eps max|eigenvalue| are dropped.  The rare fallback of `compute_stats` when R is not
        numerically positive definite (kernel_class.py:745-751); the eigen-decomposition is the vendor solver's
        (hipSOLVER through torch.linalg.eigh) and runs on the device like everything else."""
        s, u = torch.linalg.eigh(K)
        cut = s.abs().max() * K.shape[0] * torch.finfo(K.dtype).eps
        inv = torch.where(s.abs() > cut, 1.0 / s, torch.zeros_like(s))
        return (u * inv) @ u.T

    def post_het(self, K, scale, gamma_eff, y_eff, sd):
        """One draw of the mean latent of a heteroskedastic Gaussian likelihood from its exact conditional posterior
        (Hetero.post_het1 / post_het2, likelihood_class.py:184-243) with v = scale K:
            f = K alpha + u,   (K + diag(gamma_eff / scale)) alpha = y_eff - u - w,
            u = chol(v) sd[:, 0],   w = sqrt(gamma_eff) sd[:, 1].
        K: (n, n) full kernel matrix (device); gamma_eff, y_eff: (n,) device; sd: (n, 2) device.  Two factorisations:
        one for u, one (with the right-hand side riding along, inverse in the same sweep) for alpha."""
        n = K.shape[0]
        Np = self.padded_dim(n)
        A = self.workspace(('hetA', n), Np * Np * 8).view(torch.float64).view(Np, Np)
        T = self.workspace(('hetT', n), Np * Np * 8).view(torch.float64).view(Np, Np)
        S = self.workspace(('hetS', n), Np * Np * 8).view(torch.float64).view(Np, Np)
        work = self.potrf_workspace(n, 1)
        A.zero_()
        A[:n, :n] = K
        _, info1 = self.potrf(n, A, work=work)
        u = self.trmv_lower(n, A, [float(scale)], sd[:, 0].contiguous())[0]
        w = torch.sqrt(gamma_eff) * sd[:, 1]
        A.zero_()
        A[:n, :n] = K
        A.diagonal()[:n] += gamma_eff / float(scale)
        A[n, :n] = y_eff - u - w
        _, info2 = self.potrf_inv(n, A, T, S, work=work)
        alpha = -S[n, :n]
        f = self.gemv(K, alpha.contiguous()) + u
        bad = int(self.fetch(info1)[0]) or int(self.fetch(info2)[0])
        if bad:
            raise_not_pd(bad)
        return f

    def fetch(self, t):
        """Device tensor -> numpy array through the library's pinned staging buffer (one stream sync; cheaper than
        torch's .cpu() for the few bytes a sampler / optimiser step returns)."""
        t = t.contiguous()
        out = np.empty(tuple(t.shape), dtype={torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64}[t.dtype])
        self._chk(self._enter() or lib.dgpamd_fetch(self.h, _dp(t), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    MAILBOXES = 8
    # who uses which mailbox (the ranges must not overlap: a slot that still holds a result refuses the next post):
    MSTEP_SLOTS = 2      # 0 .. MSTEP_SLOTS-1: the lock-step M-step's optimiser groups in flight (mstep.py); kernel.ord_nn's default is 0,
                         # outside an M-step
    DETACH_SLOT0 = 2     # DETACH_SLOT0 + l: the imputer's deferred detach posts hidden layer l's latents (imputation.py)
    assert MSTEP_SLOTS <= DETACH_SLOT0 < MAILBOXES

    def use_dist_reduce(self):
        """Install dgp_amd.dist's all-reduce as the context's reduce hook (dgpamd_set_reduce_hook): the queued I-step of a model
        whose Vecchia likelihood rows are split over ranks sums every batch's partial sums over the ranks on the stream,
        between the row launch and the accept / shrink decision.  Idempotent."""
        if getattr(self, '_reduce_cb', None) is not None:
            return
        from . import dist as ddist
        HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)

        def hook(user, ptr, count):
            try:
                t = self._view_of(int(ptr), int(count))
                with torch.cuda.stream(self._torch_stream):
                    ddist.allreduce_sum(t)
                return 0
            except BaseException as exc:   # noqa: BLE001  (never unwind through the C frames: the call fails, _chk re-raises)
                self._reduce_exc = exc
                return 1
        self._reduce_cb = HOOK(hook)   # (kept alive with the engine)
        self._chk(self._enter() or lib.dgpamd_set_reduce_hook(self.h, C.cast(self._reduce_cb, C.c_void_p), None))

    def _view_of(self, ptr, count):
        """float64 view of `count` doubles at device address `ptr` inside one of the engine's workspaces."""
        for t in self._ws.values():
            base = t.data_ptr()
            if base <= ptr and ptr + 8 * count <= base + t.numel() * t.element_size():
                off = ptr - base
                return t.view(torch.uint8)[off:off + 8 * count].view(torch.float64)
        raise DgpAmdError('reduce hook: the buffer is not inside a workspace of this engine')

    def post(self, t, slot):
        """First half of fetch(): queue the copy of device tensor t into mailbox `slot` (0..7) behind everything queued so
        far and return a token at once (dgpamd_post).  Launches made afterwards do not delay it."""
        t = t.contiguous()
        self._chk(self._enter() or lib.dgpamd_post(self.h, _dp(t), t.numel() * t.element_size(), int(slot)))
        return (int(slot), tuple(t.shape), t.dtype, t)   # (t is kept alive until collect())

    def collect(self, token):
        """Second half: wait for THAT copy (not for the stream) and return the numpy array (dgpamd_collect)."""
        slot, shape, dtype, _ = token
        out = np.empty(shape, dtype={torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64}[dtype])
        self._chk(self._enter() or lib.dgpamd_collect(self.h, slot, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def discard(self, token):
        """Wait for a posted copy and drop it (the mailbox is free again)."""
        self._chk(self._enter() or lib.dgpamd_collect(self.h, token[0], None, 0))

    def fetch_ll_info(self, ll, info):
        """(ll float64 (B,), info int32 (B,)) device tensors -> two numpy arrays with ONE synchronisation."""
        B = ll.numel()
        buf = np.empty(12 * B, dtype=np.uint8)
        self._chk(self._enter() or lib.dgpamd_fetch2(self.h, _dp(ll), 8 * B, _dp(info), 4 * B, buf.ctypes.data_as(C.c_void_p)))
        return buf[:8 * B].view(np.float64), buf[8 * B:].view(np.int32)

    def linkgp_cells(self, kind, W_host, Wg, Rinv, ry):
        """Prediction statistics of a linked Matern-2.5 node with the training points grouped by cells (cell_order): the
        pair kernel then needs one record product instead of two for about half of its work (linkgp_Jsep_kernel's order
        classes).  W_host: (n x Dw) numpy inputs; Wg (n x Dz device or None), Rinv (ld x ld device, valid [:n, :n]), ry (n
        device) in the same row order.  Returns dict(W, Wg, Rinv, ry, pos) -- pos[i] = row of training point i in the new
        order (int32, what a leave-one-out call passes as `drop`) -- or None for kernels whose pair phase has no classes.
        The predictions are the same sums over all pairs (functions.py:421-494) in another order."""
        if kind != 'matern2.5' or os.environ.get('DGPAMD_LINK_CELLS', '1') == '0':
            return None
        n = len(W_host)
        p = cell_order(W_host)
        pd = torch.as_tensor(p, device=Rinv.device)
        pos = np.empty(n, dtype=np.int32)
        pos[p] = np.arange(n, dtype=np.int32)
        Rp = torch.zeros_like(Rinv)
        Rp[:n, :n] = Rinv[:n, :n][pd][:, pd]
        return dict(W=self.tensor(np.ascontiguousarray(np.asarray(W_host, dtype=float)[p])), Wg=None if Wg is None else Wg[pd].contiguous(),
                    Rinv=Rp, ry=ry[pd].contiguous(), pos=torch.as_tensor(pos, device=Rinv.device))

    def lik_loglik(self, lik, colmap, FP):
        """A likelihood node's log-likelihood of every candidate block FP[b] (B x n x M) -> device tensor (B,)
        (dgpamd_lik_loglik; imputation.py:71-78,91-106 -> <likelihood>.llik()).  lik: dict(kind, y, rep, classes, par) with
        device tensors y (observations) and rep (int64 latent row per observation, or None)."""
        B, n, M = FP.shape
        nd = _lib.Node()
        keep = _fill_lik_node(nd, colmap, lik, M)
        work = self.workspace(('likW', B), int(lib.dgpamd_lik_workspace(B)))
        out = self.empty(B)
        self._chk(self._enter() or lib.dgpamd_lik_loglik(self.h, C.byref(nd), n, M, _dp(FP), n * M, B, _dp(work), _dp(out)))
        del keep
        return out

    def ess_queue_plan(self, n, M, nodes, batch):
        return _EssQueue(self, n, M, nodes, batch)

    def ess_plan(self, n, M, kind, colmap, Xglob, length, nugget, W, y, batch):
        """Static arguments of dgpamd_ess_update for one upper GP node (see _EssPlan.run)."""
        return _EssPlan(self, n, M, kind, colmap, Xglob, length, nugget, W, y, batch)

    def llik_plan(self, n, specs):
        """Prepare the static part of dgpamd_llik_batch for a fixed set of GP nodes of size n.  specs: list of dicts
        kind, Xloc (n x ldloc tensor), Xglob (tensor or None), nlen, nugget_est, W (tensor or None), y (tensor).  Returns a
        plan whose per-call inputs are only the hyper-parameters (plan.set) -- no allocation, one C call per round."""
        return _LlikPlan(self, n, specs)

    def gp_predict(self, kind, x, Wtr, length, Rinv, ldr, ry, scale, nugget, mean=None, var=None):
        """ry: (n,) -> mean (M,) ; ry: (S, n) -> mean (S, M).  var is (M,) either way."""
        M, D = x.shape
        n = Wtr.shape[0]
        length = _f64(length)
        nry = 1 if ry.dim() == 1 else ry.shape[0]
        work = self.workspace(('gp', n, M), lib.dgpamd_gp_workspace(n, M))
        if mean is None:
            mean = self.empty(M) if ry.dim() == 1 else self.empty(nry, M)
        if var is None:
            var = self.empty(M)
        self._chk(self._enter() or lib.dgpamd_gp_predict(self.h, KIND[kind], n, M, D, _dp(x), _dp(Wtr), _hp(length), len(length), _dp(Rinv),
                                        ldr, _dp(ry), nry, float(scale), float(nugget), _dp(mean), _dp(var), _dp(work)))
        return mean, var

    def linkgp_predict(self, kind, m, v, z, Wtr, Wg, length, Rinv, ldr, ry, scale, nugget, mean=None, var=None,
                       drop=None):
        """drop (M,) int32 device tensor: leave training point drop[t] out of test point t's conditioning set."""
        M, Dw = m.shape
        Dz = 0 if z is None else z.shape[1]
        n = Wtr.shape[0]
        length = _f64(length)
        work = self.workspace(('link', n, M, Dw), lib.dgpamd_linkgp_workspace(n, M, Dw))
        if mean is None:
            mean = self.empty(M)
        if var is None:
            var = self.empty(M)
        if drop is not None:
            assert drop.dtype == torch.int32 and drop.numel() == M and drop.is_contiguous()
            self._chk(self._enter() or lib.dgpamd_linkgp_loo(self.h, KIND[kind], n, M, Dw, Dz, _dp(m), _dp(v), _dp(z), _dp(Wtr), _dp(Wg),
                                            _hp(length), len(length), _dp(Rinv), ldr, _dp(ry), _dp(drop), float(scale),
                                            float(nugget), _dp(mean), _dp(var), _dp(work)))
            return mean, var
        self._chk(self._enter() or lib.dgpamd_linkgp_predict(self.h, KIND[kind], n, M, Dw, Dz, _dp(m), _dp(v), _dp(z), _dp(Wtr), _dp(Wg),
                                            _hp(length), len(length), _dp(Rinv), ldr, _dp(ry), float(scale),
                                            float(nugget), _dp(mean), _dp(var), _dp(work)))
        return mean, var

    def moments_accumulate(self, mu, var, sum_mu, sum_m2):
        self._chk(self._enter() or lib.dgpamd_moments_accumulate(self.h, mu.numel(), _dp(mu), _dp(var), _dp(sum_mu), _dp(sum_m2)))

    def moments_finalize(self, S, sum_mu, sum_m2):
        self._chk(self._enter() or lib.dgpamd_moments_finalize(self.h, sum_mu.numel(), float(S), _dp(sum_mu), _dp(sum_m2)))

    # --------------------------------------------------------------- vecchia
    def nn_ordered(self, x, m):
        n, D = x.shape
        m = min(m, n - 1)
        out = self.empty(n, m + 1, dtype=torch.int64)
        self._chk(self._enter() or lib.dgpamd_nn_ordered(self.h, n, D, _dp(x), m, _dp(out)))
        return out

    def nn_query(self, q, x, m):
        M, D = q.shape
        n = x.shape[0]
        m = min(m, n)
        out = self.empty(M, m, dtype=torch.int64)
        self._chk(self._enter() or lib.dgpamd_nn_query(self.h, M, n, D, _dp(q), _dp(x), m, _dp(out)))
        return out

    def vecchia_llik(self, kind, X, y, NN, length, nugget, nugget_diag):
        """(quad, logdet) summed over the rows of NN.  NN may be a block of rows of the neighbour array (the rows a rank
        owns, dist.vecchia_rows): X, y and nugget_diag are indexed through its entries, so they stay whole."""
        n, D = NN.shape[0], X.shape[1]
        length = _f64(length)
        out = self.empty(2)
        self._chk(self._enter() or lib.dgpamd_vecchia_llik(self.h, KIND[kind], n, D, NN.shape[1] - 1, _dp(X), _dp(y), _dp(NN), _hp(length),
                                          len(length), float(nugget), _dp(nugget_diag), _dp(out)))
        return out

    def vecchia_llik_batch(self, kind, Xall, y, NN, length, nugget, nugget_diag):
        """vecchia_llik for the B input sets Xall (B, n_points, D) at once: (B, 2) sums, one row launch + one reduction."""
        B, D = Xall.shape[0], Xall.shape[2]
        length = _f64(length)
        out = self.empty(B, 2)
        self._chk(self._enter() or lib.dgpamd_vecchia_llik_batch(self.h, KIND[kind], NN.shape[0], D, NN.shape[1] - 1, _dp(Xall),
                                                                 Xall.stride(0), B, _dp(y), _dp(NN), _hp(length), len(length),
                                                                 float(nugget), _dp(nugget_diag), _dp(out)))
        return out

    def vecchia_nllik(self, kind, X, y, NN, length, nugget, nugget_diag, nugget_est):
        n, D = NN.shape[0], X.shape[1]   # (rows of NN: possibly one rank's block, see vecchia_llik)
        length = _f64(length)
        P = (1 if len(length) == 1 else D) + (1 if nugget_est else 0)
        out = self.empty(2 + 2 * P)
        self._chk(self._enter() or lib.dgpamd_vecchia_nllik(self.h, KIND[kind], n, D, NN.shape[1] - 1, _dp(X), _dp(y), _dp(NN),
                                           _hp(length), len(length), float(nugget), _dp(nugget_diag),
                                           1 if nugget_est else 0, _dp(out)))
        return out, P

    def vecchia_lmatrix(self, kind, X, NN, length, nugget):
        n, D = X.shape
        length = _f64(length)
        out = self.empty(n, NN.shape[1])
        self._chk(self._enter() or lib.dgpamd_vecchia_lmatrix(self.h, KIND[kind], n, D, NN.shape[1] - 1, _dp(X), _dp(NN), _hp(length),
                                             len(length), float(nugget), _dp(out)))
        return out

    def vecchia_post_het(self, kind, X_ord, impNN, scale, length, gamma, y, z):
        """Draw of the mean latent from its exact conditional posterior under a Hetero likelihood, Vecchia form
        (likelihood_class.py:153-182 on vecchia.U_matrix_sp :599-610), in ORDERED coordinates:
        f = -U_l^-T U_ol^T y + U_l^-T z.  X_ord (n x D), impNN (n x (m+1)) int64, gamma / y / z (n), all on the device."""
        import torch
        n, D = X_ord.shape
        mp1 = impNN.shape[1]
        length = _f64(length)
        Lrows, t = self.empty(n, mp1), self.empty(n)
        NNl = torch.empty((n, mp1), dtype=torch.int64, device=X_ord.device)
        info = torch.empty(1, dtype=torch.int32, device=X_ord.device)
        self._chk(self._enter() or lib.dgpamd_vecchia_het_rows(self.h, KIND[kind], n, D, mp1 - 1, _dp(X_ord), _dp(impNN), _hp(length), len(length),
                                              float(scale), _dp(gamma), _dp(y), _dp(Lrows), _dp(NNl), _dp(t), _dp(info)))
        bad = int(info.item())
        if bad:
            raise np.linalg.LinAlgError('Vecchia block of row %d is not positive definite' % (bad - 1))
        return self.vecchia_spsolve(Lrows, NNl, 1.0, z) - self.vecchia_spsolve(Lrows, NNl, 1.0, t)

    def vecchia_spsolve(self, Lmat, NN, inv_sqrt_scale, b):
        n = Lmat.shape[0]
        out = self.empty(n)
        self._chk(self._enter() or lib.dgpamd_vecchia_spsolve(self.h, n, NN.shape[1] - 1, _dp(Lmat), _dp(NN), float(inv_sqrt_scale), _dp(b),
                                             _dp(out)))
        return out

    def vecchia_spsolve_batch(self, Lmat, NN, inv_sqrt_scale, b):
        """Lmat, NN: (nmat, n, m+1); inv_sqrt_scale: nmat floats; b: (nmat, nrhs, n) -> x of the same shape."""
        nmat, n, mp1 = Lmat.shape
        nrhs = b.shape[1]
        out = self.empty(nmat, nrhs, n)
        sc = self.tensor(np.asarray(inv_sqrt_scale, dtype=np.float64))
        self._chk(self._enter() or lib.dgpamd_vecchia_spsolve_batch(self.h, n, mp1 - 1, nmat, nrhs, _dp(Lmat), _dp(NN), _dp(sc), _dp(b), _dp(out)))
        return out

    def vecchia_levels(self, NN):
        """Level schedule of the sparse forward substitution for the neighbour arrays NN (nmat, n, m+1) or (n, m+1): an int32
        device tensor for vecchia_spsolve_levels; depends on NN only (build once per ordering)."""
        if NN.dim() == 2:
            NN = NN.unsqueeze(0)
        nmat, n, mp1 = NN.shape
        sched = torch.empty(int(lib.dgpamd_vecchia_levels_bytes(n, nmat)) // 4, dtype=torch.int32, device=self.device)
        self._chk(self._enter() or lib.dgpamd_vecchia_levels(self.h, n, mp1 - 1, nmat, _dp(NN.contiguous()), _dp(sched)))
        return sched

    def vecchia_spsolve_levels(self, Lmat, NN, inv_sqrt_scale, b, sched):
        """vecchia_spsolve_batch on a level schedule (vecchia_levels of the same NN)."""
        nmat, n, mp1 = Lmat.shape
        nrhs = b.shape[1]
        out = self.empty(nmat, nrhs, n)
        sc = self.tensor(np.asarray(inv_sqrt_scale, dtype=np.float64))
        self._chk(self._enter() or lib.dgpamd_vecchia_spsolve_levels(self.h, n, mp1 - 1, nmat, nrhs, _dp(Lmat), _dp(NN), _dp(sc), _dp(b), _dp(out),
                                                                   _dp(sched)))
        return out

    def vecchia_gp(self, kind, x, w, NN, y, scale, length, nugget, nugget_diag):
        M, D = x.shape
        length = _f64(length)
        mean, var = self.empty(M), self.empty(M)
        self._chk(self._enter() or lib.dgpamd_vecchia_gp(self.h, KIND[kind], M, w.shape[0], D, NN.shape[1], _dp(x), _dp(w), _dp(NN), _dp(y),
                                        float(scale), _hp(length), len(length), float(nugget), _dp(nugget_diag),
                                        _dp(mean), _dp(var)))
        return mean, var

    def vecchia_linkgp(self, kind, m, v, z, w1, wg, NN, y, scale, length, nugget, nugget_diag):
        M, Dw = m.shape
        Dz = 0 if z is None else z.shape[1]
        length = _f64(length)
        mean, var = self.empty(M), self.empty(M)
        self._chk(self._enter() or lib.dgpamd_vecchia_linkgp(self.h, KIND[kind], M, w1.shape[0], Dw, Dz, NN.shape[1], _dp(m), _dp(v), _dp(z),
                                            _dp(w1), _dp(wg), _dp(NN), _dp(y), float(scale), _hp(length), len(length),
                                            float(nugget), _dp(nugget_diag), _dp(mean), _dp(var)))
        return mean, var


_default = {}


def default_engine(device=None):
    """Process-wide engine for `device` (LOCAL_RANK-aware default)."""
    import os
    if device is None:
        device = int(os.environ.get('LOCAL_RANK', '0'))
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]


class _LlikPlan:
    """Arguments of dgpamd_llik_batch kept alive between the rounds of a lock-step M-step."""

    def __init__(self, eng, n, specs):
        self.e, self.n, self.B = eng, int(n), len(specs)
        B, Np = self.B, eng.padded_dim(n)
        self.keep = specs                      # tensors stay referenced
        self.nodes = (_lib.Node * B)()
        self.lengths = []
        self.P = []
        for b, sp in enumerate(specs):
            nd = self.nodes[b]
            Xl, Xg = sp['Xloc'], sp['Xglob']
            nd.kind = KIND[sp['kind']]
            nd.Dl, nd.Dg = Xl.shape[1], 0 if Xg is None else Xg.shape[1]
            nd.nlen, nd.nugget_est, nd.ldloc = int(sp['nlen']), 1 if sp['nugget_est'] else 0, Xl.shape[1]
            nd.Xloc, nd.colmap, nd.Xglob = Xl.data_ptr(), None, None if Xg is None else Xg.data_ptr()
            length = np.ones(nd.nlen)
            self.lengths.append(length)
            nd.length = length.ctypes.data
            nd.W = None if sp['W'] is None else sp['W'].data_ptr()
            nd.y = sp['y'].data_ptr()
            self.P.append((1 if nd.nlen == 1 else nd.Dl + nd.Dg) + nd.nugget_est)
        self.stride_out = 3 + 2 * max(self.P)
        self.A = eng.workspace(('mstepA', n), B * Np * Np * 8)
        self.Ainv = eng.workspace(('mstepAinv', n), B * Np * Np * 8)
        self.T = eng.workspace(('mstepT', n), B * Np * Np * 8)
        self.work = eng.potrf_workspace(n, B)
        self.gwork = eng.workspace(('gradB', n, max(self.P), B), B * lib.dgpamd_grad_workspace(n, max(self.P)))
        self.dev_out = eng.empty(B * (self.stride_out + 2))
        self.host = np.zeros((B, self.stride_out))
        self.stride_a = Np * Np

    def set(self, b, length, nugget):
        self.lengths[b][:] = length
        self.nodes[b].nugget = float(nugget)

    def factor_view(self, r):
        """The augmented buffer of row r of the LAST run() as an (Np, Np) float64 tensor: after the sweep its lower tiles
        hold the Cholesky factor of that node's K (rows < n), exactly what dgpamd_potrf leaves."""
        Np = self.e.padded_dim(self.n)
        return self.A[r * self.stride_a * 8:(r + 1) * self.stride_a * 8].view(torch.float64).view(Np, Np)

    def launch(self, idx):
        """First half of run(): everything queued on the device (dgpamd_llik_batch_launch); wait(token) returns the results."""
        e = self.e
        if len(idx) == self.B:
            nodes, B = self.nodes, self.B
        else:
            B = len(idx)
            nodes = (_lib.Node * B)(*[self.nodes[i] for i in idx])
        e._chk(e._enter() or lib.dgpamd_llik_batch_launch(e.h, self.n, B, nodes, _dp(self.A), _dp(self.T), _dp(self.Ainv), self.stride_a, _dp(self.work),
                                                          _dp(self.gwork), _dp(self.dev_out), self.stride_out))
        return (list(idx), nodes)   # (the node array stays alive until the wait)

    def wait(self, token):
        idx = token[0]
        e = self.e
        host = self.host[:len(idx)]
        e._chk(e._enter() or lib.dgpamd_llik_batch_wait(e.h, host.ctypes.data_as(C.c_void_p)))
        out = {}
        for r, i in enumerate(idx):
            P = self.P[i]
            out[i] = np.concatenate((host[r, :2], host[r, 3:3 + 2 * P], host[r, 2:3]))
        return out

    def run(self, idx):
        """Evaluate the nodes listed in idx (positions in the plan); returns {position: host vector
        [logdet, y'K^-1y, tr.., quad.., info]} in kernel._llik_device's layout."""
        e = self.e
        if len(idx) == self.B:
            nodes, B = self.nodes, self.B
        else:
            B = len(idx)
            nodes = (_lib.Node * B)(*[self.nodes[i] for i in idx])
        host = self.host[:B]
        e._chk(e._enter() or lib.dgpamd_llik_batch(e.h, self.n, B, nodes, _dp(self.A), _dp(self.T), _dp(self.Ainv), self.stride_a, _dp(self.work),
                                     _dp(self.gwork), _dp(self.dev_out), host.ctypes.data_as(C.c_void_p), self.stride_out))
        out = {}
        for r, i in enumerate(idx):
            P = self.P[i]
            out[i] = np.concatenate((host[r, :2], host[r, 3:3 + 2 * P], host[r, 2:3]))
        return out


def cell_order(W, leaf=16, block=64):
    """A permutation of the rows of W (n x D training inputs of a linked node) that groups them by cells: recursive
    splits at an order statistic of one coordinate (the coordinates in turn), every part a multiple of `block` rows while it
    is larger than a block and of `leaf` rows below, down to `leaf` rows.  Rows 16 a .. 16 a + 15 of the permuted array are
    then a cell, and two cells that were separated by a split in coordinate k satisfy max x_k <= min x_k one way round.  The
    Matern pair kernel (linkgp_Jsep_kernel) needs one of its two record products for such a (sub-tile, coordinate) and both
    for the others; any order gives the same predictions (functions.py:453-494 sums over all pairs)."""
    W = np.asarray(W, dtype=float)
    n, D = W.shape
    out = []
    stack = [(np.arange(n), 0)]
    while stack:
        idx, depth = stack.pop()
        m = len(idx)
        if m <= leaf:
            out.append(idx)
            continue
        unit = block if m > block else leaf
        left = unit * max(1, int(round(m / (2.0 * unit))))
        if left >= m:
            left = m - (m % unit or unit)
        k = depth % D
        o = idx[np.argsort(W[idx, k], kind='stable')]
        stack.append((o[left:], depth + 1))   # (popped after the left part: the output keeps left before right)
        stack.append((o[:left], depth + 1))
    return np.concatenate(out)


LIK_KIND = {'Poisson': 1, 'NegBin': 2, 'ZIP': 3, 'ZINB': 4, 'logit': 5, 'probit': 6, 'robustmax': 7, 'softmax': 8}


def _fill_lik_node(nd, colmap, lik, M):
    """dgpamd_node fields of a likelihood node (include/dgp_amd.h); returns what must stay alive with the struct."""
    colmap = np.ascontiguousarray(np.asarray(colmap, dtype=np.int32))
    nd.kind, nd.Dl, nd.Dg, nd.nlen, nd.ldloc = 0, len(colmap), 0, 0, M
    nd.colmap = colmap.ctypes.data
    nd.y = lik['y'].data_ptr()
    nd.lik_kind, nd.lik_classes = LIK_KIND[lik['kind']], int(lik.get('classes') or 0)
    nd.lik_nobs = int(lik['y'].numel())
    nd.lik_rep = None if lik.get('rep') is None else lik['rep'].data_ptr()
    nd.lik_par = float(lik.get('par') or 0.0)
    return (colmap, lik['y'], lik.get('rep'))


class _EssQueue:
    """Several elliptical-slice updates of one latent block queued without host synchronisation (dgpamd_ess_queue):
    node structs, scratch and the device state are kept alive here; fetch() is the one synchronisation.  Several queues
    (the layers of a deeper model) can share one device state and one uploaded uniform stream: share_with()."""
    STATE = 16
    FIELDS = ('theta', 'lo', 'hi', 'pending', 'cursor', 'status', 'info', 'll', 'log_y', 'proposals', 'batches', 'updates')

    def __init__(self, eng, n, M, nodes, batch):
        """nodes: list of dicts {kind, colmap, Xglob, length, nugget, W, y} -- the GP nodes of the layer above; a Vecchia
        node adds vecch = dict(ord, nn, nd, y): device tensors (ordering, neighbour array, ordered nugget weights, ordered
        outputs; kernel_class.py:494-509)."""
        self.e, self.n, self.M, self.batch = eng, int(n), int(M), int(batch)
        Np = eng.padded_dim(n)
        self.keep = []
        arr = (_lib.Node * len(nodes))()
        dense, Dv = False, 0
        nodes = list(nodes)
        for i, d in enumerate(nodes):
            if d.get('lik') is not None:   # a likelihood node: dict(kind, y, rep, classes, par) + colmap
                self.keep.append(_fill_lik_node(arr[i], d['colmap'], d['lik'], M))
                continue
            colmap = np.ascontiguousarray(np.asarray(d['colmap'], dtype=np.int32))
            length = np.ascontiguousarray(np.asarray(d['length'], dtype=np.float64))
            self.keep += [colmap, length, d['Xglob'], d['W'], d['y'], d.get('vecch')]
            nd = arr[i]
            nd.kind, nd.Dl, nd.Dg = KIND[d['kind']], len(colmap), 0 if d['Xglob'] is None else d['Xglob'].shape[1]
            nd.nlen, nd.nugget_est, nd.ldloc = len(length), 0, M
            nd.Xloc, nd.colmap = None, colmap.ctypes.data
            nd.Xglob = None if d['Xglob'] is None else d['Xglob'].data_ptr()
            nd.length, nd.nugget = length.ctypes.data, float(d['nugget'])
            nd.W = None if d['W'] is None else d['W'].data_ptr()
            nd.y = d['y'].data_ptr()
            v = d.get('vecch')
            if v is not None:
                nd.vecch_ord, nd.vecch_nn = v['ord'].data_ptr(), v['nn'].data_ptr()
                nd.vecch_nd, nd.vecch_y = v['nd'].data_ptr(), v['y'].data_ptr()
                nd.vecch_m = int(v['nn'].shape[1]) - 1
                if v.get('rows') is not None:   # this rank's block of rows (dist.split_training(rows=True)): the sums are all-reduced
                    nd.vecch_row0, nd.vecch_rows = int(v['rows'][0]), int(v['rows'][1] - v['rows'][0])
                    eng.use_dist_reduce()
                Dv = max(Dv, nd.Dl + nd.Dg)
            else:
                dense = True
        self.nodes, self.nnodes = arr, len(nodes)
        self.FP = eng.workspace(('essFP', n, M, batch), batch * n * M * 8)
        self.A = eng.workspace(('essA', n, batch), batch * Np * Np * 8) if dense else None
        self.work = eng.potrf_workspace(n, batch) if dense else None
        self.vwork = eng.workspace(('essV', n, Dv, batch), int(lib.dgpamd_ess_queue_vwork(n, Dv, batch))) if Dv else None
        self.scratch = eng.workspace(('essQ',), int(lib.dgpamd_ess_queue_scratch()))
        self.state = eng.empty(self.STATE)
        self.udev = None

    def share_with(self, other):
        """Continue `other`'s queue: the same device state and the same uploaded uniforms."""
        self.state, self.udev = other.state, other.udev

    def upload_uniforms(self, uniforms):
        e = self.e
        if self.udev is None or self.udev[0] is not uniforms:
            us = np.ascontiguousarray(np.asarray(uniforms, dtype=np.float64))
            with np.errstate(divide='ignore'):
                both = np.concatenate((us, np.log(us)))
            self.udev = (uniforms, e.tensor(both), len(us))

    def reset_state(self, cursor=0, ll=None):
        """A fresh device state (status, counters, info zero; the uniform cursor and the cached log-likelihood as given).  What
        queue(fresh=True) does first; callers that queue other work between the reset and the first update -- a deeper layer's
        prior factorisation, whose info word note_info() folds into this state -- reset here and queue with fresh=False."""
        st0 = np.zeros(self.STATE)
        st0[4], st0[7] = cursor, 0.0 if ll is None else ll
        self.state.copy_(self.e.tensor(st0))

    def queue(self, F, NU, scales, uniforms, cursor, ll, compute_ll0, batch_next, max_batches, fresh=True):
        """Queue NU.shape[0] updates (NU: (nupd, n, M) device tensor).  uniforms: the sampler's upcoming uniforms (host);
        cursor: how many of them earlier queues of this I-step have consumed.  fresh=False: the device state is the one an
        earlier queue() of this I-step left (cursor, status and counters carry on; `cursor` and `ll` are ignored)."""
        e = self.e
        self.upload_uniforms(uniforms)
        ud, nuni = self.udev[1], self.udev[2]
        if fresh:
            self.reset_state(cursor, ll)
        sc = np.ascontiguousarray(np.asarray(scales, dtype=np.float64))
        e._chk(e._enter() or lib.dgpamd_ess_queue(e.h, self.n, self.M, _dp(F), _dp(NU), int(NU.shape[0]), C.cast(self.nodes, C.c_void_p),
                                    sc.ctypes.data_as(C.c_void_p), self.nnodes, _dp(self.state), _dp(ud), _dp(ud[nuni:]), nuni,
                                    self.batch, int(batch_next) if batch_next else self.batch, int(max_batches),
                                    int(compute_ll0), _dp(self.FP), _dp(self.A), _dp(self.work), _dp(self.scratch),
                                    _dp(self.vwork)))

    def resume_state(self, st, ll=None):
        """The device state for a queue that CONTINUES the update an earlier one left open (status 3 / 1): angle, bracket, pending flag and threshold as
        fetched (`st`: fetch()'s dict), status and counters zero, the cursor at the start of the uniforms the next queue() uploads."""
        st0 = np.zeros(self.STATE)
        st0[0], st0[1], st0[2], st0[3] = st['theta'], st['lo'], st['hi'], st['pending']
        st0[7] = st['ll'] if ll is None else ll
        st0[8] = st['log_y']
        self.state.copy_(self.e.tensor(st0))

    def note_info(self, info):
        """A factorisation queued between two updates (a deeper layer's prior factors): non-zero info stops the queue."""
        e = self.e
        e._chk(e._enter() or lib.dgpamd_ess_queue_note_info(e.h, _dp(self.state), _dp(info), int(info.numel())))

    def fetch(self):
        """The ONE synchronisation: the device state as a dict."""
        v = self.e.fetch(self.state)
        return dict(zip(self.FIELDS, v[:len(self.FIELDS)]))


class _EssPlan:
    """One elliptical-slice update per call: dgpamd_ess_update with its scratch buffers kept alive."""

    def __init__(self, eng, n, M, kind, colmap, Xglob, length, nugget, W, y, batch):
        self.e, self.n, self.M, self.batch = eng, int(n), int(M), int(batch)
        Np = eng.padded_dim(n)
        self.keep = (Xglob, W, y)
        self.colmap = np.ascontiguousarray(np.asarray(colmap, dtype=np.int32))
        self.length = np.ascontiguousarray(np.asarray(length, dtype=np.float64))
        nd = self.node = _lib.Node()
        nd.kind, nd.Dl, nd.Dg = KIND[kind], len(self.colmap), 0 if Xglob is None else Xglob.shape[1]
        nd.nlen, nd.nugget_est, nd.ldloc = len(self.length), 0, M
        nd.Xloc, nd.colmap = None, self.colmap.ctypes.data
        nd.Xglob = None if Xglob is None else Xglob.data_ptr()
        nd.length, nd.nugget = self.length.ctypes.data, float(nugget)
        nd.W = None if W is None else W.data_ptr()
        nd.y = y.data_ptr()
        self.FP = eng.workspace(('essFP', n, M, batch), batch * n * M * 8)
        self.A = eng.workspace(('essA', n, batch), batch * Np * Np * 8)
        self.work = eng.potrf_workspace(n, batch)
        self.ll = eng.empty(batch)
        self.info = eng.empty(batch, dtype=torch.int32)
        self.state = np.zeros(4)
        self.out = np.zeros(6)

    def run(self, F, NU, scale, log_y, theta, lo, hi, pending, uniforms, batch_next):
        """Returns (status, consumed, proposals, batches, ll, info, theta, lo, hi, pending)."""
        e = self.e
        us = np.ascontiguousarray(np.asarray(uniforms, dtype=np.float64))
        self.state[:] = (theta, lo, hi, 1.0 if pending else 0.0)
        e._chk(e._enter() or lib.dgpamd_ess_update(e.h, self.n, self.M, _dp(F), _dp(NU), C.byref(self.node), float(scale), float(log_y),
                                     self.state.ctypes.data_as(C.c_void_p), us.ctypes.data_as(C.c_void_p), len(us),
                                     self.batch, int(batch_next) if batch_next else self.batch, _dp(self.FP), _dp(self.A),
                                     _dp(self.work), _dp(self.ll), _dp(self.info), self.out.ctypes.data_as(C.c_void_p)))
        o, st = self.out, self.state
        return int(o[0]), int(o[1]), int(o[2]), int(o[3]), float(o[4]), int(o[5]), st[0], st[1], st[2], bool(st[3])
