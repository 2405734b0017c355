"""Elliptical-slice-sampling imputer (I-step of SI) -- device resident.

Mirrors dgpsi's `imputer` (imputation.py:6-262): `sample(burnin)`, `key_stats()`,
`update_ord_nn()`; the latent layers live in HBM for the whole call and the numpy
attributes of the nodes (`output`, `input`) are refreshed when it returns, so code
that reads them afterwards (M-step, emulator, user scripts) sees what the reference
would have left there.

What is restructured (results unchanged):
  * the proposal angles of one ESS update are a function of the uniform draws alone
    (imputation.py:81-82,115-119), so B speculative proposals are evaluated as ONE
    batched kernel-assembly + Cholesky and the first one above the threshold is
    taken -- identical to the sequential loop, including how many uniforms are consumed;
  * a first-layer node's prior factor chol(K) depends only on X and its
    hyper-parameters: it is cached until either changes (imputation.py:63 refactors
    it on every sweep);
  * the threshold log-likelihood of an unchanged state is the one computed when that
    state was accepted.
Randomness: one numpy Generator for fmvn's normals and one for the uniforms (the
reference also uses two streams: numba's for randn, numpy's global for uniform);
both can be injected for deterministic replay.
"""
import os
import numpy as np
from numpy.linalg import LinAlgError
from .ops import raise_not_pd
from . import dist as ddist
import torch

from .ops import default_engine
from .kernel_class import TrackedInputs, bind_private, peek

TWO_PI = 2.0 * np.pi
_NO_UPLOAD_AHEAD = bool(__import__('os').environ.get('DGPAMD_NO_UPLOAD_AHEAD'))   # (A/B switch of DrawStream.prefetch's early upload)


class DrawStream:
    """Normal and uniform draws with look-ahead on the uniforms (speculative proposals must
    not consume more uniforms than the sequential sampler would)."""

    def __init__(self, seed=None, z=None, u=None):
        ss = seed if isinstance(seed, np.random.SeedSequence) else np.random.SeedSequence(seed)
        a, b = ss.spawn(2)
        self._gz, self._gu = np.random.default_rng(a), np.random.default_rng(b)
        self._z = None if z is None else [np.asarray(v, dtype=float) for v in z]
        self._ubuf = [] if u is None else [float(v) for v in u]
        self._injected_u = u is not None
        self._zbuf, self._zpos, self._thread = None, 0, None

    def normal(self, n):
        if self._z is not None:
            v = self._z.pop(0)
            assert len(v) == n, 'injected normal draw has the wrong length'
            return v
        return self.normals(n)

    def normals(self, count):
        """The next `count` standard normals of the stream.  They come from a buffer that prefetch() may have filled
        on a background thread; the sequence is the one the generator yields either way."""
        self._join()
        buf = self._zbuf
        if buf is not None and len(buf) - self._zpos >= count:
            out = buf[self._zpos:self._zpos + count]
            self._zpos += count
            return out
        rest = None if buf is None else buf[self._zpos:]
        self._zbuf, self._zpos = None, 0
        return self._fill(rest, count)

    def _fill(self, rest, count):
        """`count` normals: the unread tail `rest` of the old buffer, then fresh ones -- in page-locked host memory when a
        device is there (the I-step uploads them: 35 MB per call at n = 50 000 with eight nodes, which from pageable
        memory goes through the runtime's staging chunks at ~1.4 GB/s)."""
        have = 0 if rest is None else len(rest)
        out = None
        if count >= (1 << 16):
            try:
                import torch
                if torch.cuda.is_available():
                    # two page-locked buffers in turn (the previous one may still be read by an upload in flight); BOTH are
                    # checked against this request -- after a larger request has grown one of them the other one is still
                    # the old size (it was handed out short)
                    pins = self.__dict__.setdefault('_zpin', [])
                    if len(pins) == 2:
                        pins.reverse()   # (pins[0] becomes the older one; `rest` may point into the other)
                    if len(pins) < 2:
                        pins.insert(0, torch.empty(count, dtype=torch.float64, pin_memory=True))
                    elif pins[0].numel() < count:
                        pins[0] = torch.empty(count, dtype=torch.float64, pin_memory=True)
                    out = pins[0].numpy()[:count]
                    self._zpin_cur = pins[0]
            except Exception:
                out = None
        if out is None:
            out = np.empty(count)
        if have:
            out[:have] = rest
        if count > have:
            self._gz.standard_normal(out=out[have:])
        assert len(out) == count
        return out

    def prefetch(self, count, engine=None):
        """Generate the next `count` normals on a background thread (numpy releases the GIL while filling), e.g. while
        the device is busy with the M-step; normals() then hands them out in order.  With `engine` the thread also starts
        their upload on a side stream (35 MB per I-step at n = 50 000 with eight nodes: 2 ms of an otherwise idle device at
        the start of every I-step when it waits for the copy): normals_device() hands out the device copy."""
        if self._z is not None or count <= 0:
            return
        self._join()
        have = 0 if self._zbuf is None else len(self._zbuf) - self._zpos
        if have >= count:
            return
        import threading
        rest = None if have == 0 else self._zbuf[self._zpos:].copy()

        def work():
            self.__dict__.pop('_zdev', None)
            self.__dict__.pop('_zpin_cur', None)
            self._zbuf = self._fill(rest, count)
            self._zpos = 0
            pin = self.__dict__.get('_zpin_cur')
            if engine is not None and pin is not None and not _NO_UPLOAD_AHEAD:
                try:
                    torch.cuda.set_device(engine.device)
                    side = engine.__dict__.get('_side_stream')
                    if side is None:
                        side = engine._side_stream = torch.cuda.Stream(engine.device)
                    with torch.cuda.stream(side):
                        dev = pin[:count].to(engine.device, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    self._zdev = (self._zbuf, dev, ev)
                except Exception:   # noqa: BLE001  (the host copy is still there)
                    self.__dict__.pop('_zdev', None)
        self._zbuf, self._zpos = None, 0
        self._thread = threading.Thread(target=work)   # (not a daemon: the interpreter waits for it at exit -- killed inside a HIP call it took the process down with std::terminate)
        self._thread.start()

    def normals_device(self, engine, count):
        """normals(count) as a device tensor: the copy prefetch(engine=...) started if it covers exactly these draws,
        else an upload now.  Same draws either way."""
        self._join()
        hit = self.__dict__.get('_zdev')
        buf = self._zbuf
        if hit is not None and buf is not None and hit[0] is buf and len(buf) - self._zpos >= count:
            lo = self._zpos
            self._zpos += count
            engine._torch_stream.wait_event(hit[2])
            hit[1].record_stream(engine._torch_stream)   # (allocated on the side stream, read on the engine's)
            return hit[1][lo:lo + count]
        return engine.tensor(self.normals(count))

    def __getstate__(self):
        self._join()
        st = dict(self.__dict__)
        st['_thread'] = None
        st.pop('_zpin', None)
        st.pop('_zdev', None)
        st.pop('_zpin_cur', None)
        if st.get('_zbuf') is not None:
            st['_zbuf'] = np.array(st['_zbuf'])   # (out of the page-locked buffer)
        return st

    def _join(self):
        t = self._thread
        if t is not None:
            t.join()
            self._thread = None

    def uniform_peek(self, k):
        """Up to k upcoming uniforms WITHOUT consuming them (an injected stream may hold fewer)."""
        while len(self._ubuf) < k and not self._injected_u:
            self._ubuf.extend(self._gu.random(max(k, 16)).tolist())
        return self._ubuf[:k]

    def uniform_take(self, k=1):
        out = self.uniform_peek(k)
        if len(out) < k:
            raise RuntimeError('injected uniform stream exhausted')
        del self._ubuf[:k]
        return out

    def exhausted(self):
        return (self._z is not None and len(self._z) == 0) and (self._injected_u and len(self._ubuf) == 0)


def shrink(theta, lo, hi, u):
    """One rejection step of the ESS bracket (imputation.py:115-119); numpy's uniform(lo,hi) = lo+(hi-lo)u."""
    if theta < 0.0:
        lo = theta
    else:
        hi = theta
    return lo + (hi - lo) * u, lo, hi


def speculative_angles(theta, lo, hi, us):
    """theta followed by the angles the next len(us) rejections would produce."""
    th, br = [theta], [(lo, hi)]
    for u in us:
        theta, lo, hi = shrink(theta, lo, hi, u)
        th.append(theta)
        br.append((lo, hi))
    return th, br


class imputer:
    """Args as dgpsi.imputer (imputation.py:13) plus `draws` (a DrawStream), `engine`, and
    `batch` (speculative proposals per launch)."""

    def __init__(self, all_layer, block=True, draws=None, engine=None, batch=10):
        self.all_layer = all_layer
        self.block = block
        self.draws = draws if draws is not None else DrawStream()
        self._engine = engine
        self.batch = int(batch)
        self._batch_default = int(batch) == 10   # (the queue picks its own sizes for Vecchia nodes upstairs unless the caller chose)
        self.batch_next = 6      # size of the 2nd, 3rd, ... speculative batch of an update: after a rejected batch the
                                 # bracket is narrow and acceptance is near.  10 then 6 (round 4, with the one-launch kernel's
                                 # new timings -- a batch of 10 costs 0.79 ms, of 12 0.91: tools/gpu_ess_batch_choice.py, an update
                                 # needs 7.9 proposals on average and more than 16 in 2 % of the cases): bench 34.1 against 34.7 ms
                                 # per iteration with 12 then 4 (rounds 1-3) on the same training path
        self._factor_cache = {}
        self._ess_plans = {}
        self.stats = dict(proposals=0, updates=0, batches=0)

    @property
    def engine(self):
        if self._engine is None:
            self._engine = default_engine()
        return self._engine

    def __getstate__(self):
        self.finish_detach()
        st = dict(self.__dict__)
        st['_engine'] = None
        st['_factor_cache'] = {}
        st['_ess_plans'] = {}
        for key in ('_attached', 'F', '_glob', '_yy', '_x0', '_ll_cache', '_const', '_Fh', '_vecch_dev', '_vecch_y', '_adopt', '_adopt_ll', '_given_inputs', '_sp_levels', '_lik_cache'):   # device state: rebuilt by the next sample()
            st.pop(key, None)
        return st

    # ------------------------------------------------------------------ device state
    def _dev_const(self, key, arr, private=False):
        """Device copy of a host array that rarely changes (inputs, observed outputs): uploaded again only when the host
        values differ from the ones uploaded last (the numpy attributes stay the source of truth, as in the reference).
        private: the array object is one nobody outside the library holds (TrackedInputs._private) -- as long as the node
        still binds THAT object its values cannot have changed, and the comparison by value (0.3 ms per 3-MB array, a dozen
        per iteration at n = 50 000) is skipped; every other array is compared on every call and stays writable."""
        cache = self.__dict__.setdefault('_const', {})
        hit = cache.get(key)
        if self._const_same(key, arr, private):
            return hit[1]
        if hit is not None:
            self._inputs_changed = True   # (values that were uploaded before have changed: nothing computed from them may be reused)
        src = arr
        arr = np.ascontiguousarray(arr, dtype=float)
        t = self.engine.tensor(arr)
        owner = src if (private and isinstance(src, np.ndarray)) else None
        cache[key] = (arr if (arr is not src and not np.shares_memory(arr, src)) else arr.copy(), t, owner)
        return t

    def _const_same(self, key, arr, private=False):
        """Does the host array `arr` hold the values last uploaded under `key`?"""
        hit = self.__dict__.get('_const', {}).get(key)
        if hit is None:
            return False
        if private and len(hit) > 2 and hit[2] is not None and arr is hit[2]:
            return True
        a = np.asarray(arr, dtype=float)
        return hit[0].shape == a.shape and np.array_equal(hit[0], a)

    def _attach(self, trusted=False):
        """Device state for a sample() call.  trusted (dgp.train, from its second iteration on): the nodes' arrays are the ones
        this imputer wrote at the end of the previous call and nobody has touched them since -- the device copies are kept as
        they are, nothing is compared (0.4 ms per call at n = 50 000, with the device idle)."""
        if trusted and self.__dict__.get('F') is not None and self.__dict__.get('_attached') and not self.__dict__.get('_inputs_changed'):
            self._Fh = {}
            self._ll_cache = {}
            return
        self._attached = False
        L = len(self.all_layer)
        # latents: what _detach wrote to the nodes is still on the device unless somebody changed the numpy side since
        Fh = self.__dict__.get('_Fh', {})
        old = self.__dict__.get('F', None)
        F = []
        for l in range(L - 1):
            cols = [np.asarray(nd.output, dtype=float).reshape(-1) for nd in self.all_layer[l]]
            # (column by column against what _detach fetched: no (n, M) copy when nothing changed -- 2 ms at n = 50 000)
            keep = old is not None and l < len(old) and l in Fh and Fh[l].shape == (len(cols[0]), len(cols)) and \
                all(np.array_equal(Fh[l][:, k], c) for k, c in enumerate(cols))
            if not keep:
                self.__dict__.pop('_adopt_ll', None)   # (the latents are not the ones the M-step saw)
            F.append(old[l] if keep else self.engine.tensor(np.stack(cols, 1)))
        self.F = F
        self._Fh = {}
        self._glob = {}
        self._yy = {}
        self._x0 = {}
        for l in range(L):
            for k, nd in enumerate(self.all_layer[l]):
                if nd.type != 'gp':
                    continue
                self._glob[(l, k)] = None if nd._global_input is None else self._dev_const(('g', l, k), nd._global_input, nd._private('global_input'))
                if l == 0:
                    self._x0[k] = self._dev_const(('x', k), nd._input, nd._private('input'))
                if l == L - 1:
                    self._yy[k] = self._dev_const(('y', k), np.asarray(nd.output, dtype=float).reshape(-1))
        self._ll_cache = {}
        if self.__dict__.pop('_inputs_changed', False):   # an input was edited (in place or re-bound) since the last call
            self._factor_cache = {}
            self.__dict__.pop('_adopt_ll', None)
            self.__dict__.pop('_adopt', None)
        self._attached = True

    def finish_detach(self):
        """The deferred half of sample(detach=False): bring the nodes' numpy attributes up to date."""
        if self.__dict__.pop('_detach_pending', False):
            self._detach()

    def _detach(self, defer=False):
        """Refresh the numpy attributes the reference's sampler mutates (imputation.py:94,109)."""
        L = len(self.all_layer)
        if defer:
            self._detach_pending = True
            # the latents start their way to the host NOW, ahead of whatever the caller queues next (the M-step's first
            # evaluations): finish_detach() waits for these copies only
            self._F_posted = {l: self.engine.post(self.F[l], self.engine.DETACH_SLOT0 + l) for l in range(L - 1)} if L - 1 <= self.engine.MAILBOXES - self.engine.DETACH_SLOT0 else {}
            return
        self.__dict__.pop('_detach_pending', None)
        posted = self.__dict__.pop('_F_posted', {})
        self._Fh = {}
        for l in range(L - 1):
            # (pinned staging: ~10x cheaper than torch's pageable .cpu() for 80 KB)
            Fh = self.engine.collect(posted[l]) if l in posted else self.engine.fetch(self.F[l])
            self._Fh[l] = Fh
            for k, nd in enumerate(self.all_layer[l]):
                nd.output[:, 0] = Fh[:, k]
            given = self.__dict__.setdefault('_given_inputs', {})
            for nd in self.all_layer[l + 1]:
                if nd.rep is not None and nd.type == 'likelihood':
                    new_in = Fh[nd.rep, :][:, nd.input_dim]
                else:
                    new_in = Fh[:, nd.input_dim]
                if isinstance(nd, TrackedInputs):
                    nd._input = new_in           # a private copy: stage_for_mstep recognises it by identity until it is handed out
                    given[id(nd)] = new_in
                else:
                    nd.input = new_in            # (a user plugin node: a plain attribute)

    def adopt_from_mstep(self, nd, Aslot, host):
        """Called by the lock-step M-step when a node's optimiser has ended, with the factored buffer and the host results of
        its last evaluation -- which is at the node's final hyper-parameters.  For a dense first-layer node that factor IS
        the prior factor the next I-step draws with (K depends on the fixed inputs and the hyper-parameters only): it is
        copied into the factor cache (33 MB on the device) instead of being recomputed.  For a node of the last layer of a
        two-layer model, -0.5 (n log s + logdet + y'K^-1y / s) is the log-likelihood of the current latents under the new
        hyper-parameters: the base of the next I-step's first slice threshold.  Both save the I-step its opening
        factorisation launch."""
        if nd.type != 'gp' or nd.vecch or nd.rep is not None or self.__dict__.get('F') is None:
            return
        L = len(self.all_layer)
        for l, layer in enumerate(self.all_layer):
            for k, cand in enumerate(layer):
                if cand is not nd:
                    continue
                e = self.engine
                n = len(nd.output)
                if l == 0:
                    M = len(layer)
                    Np = e.padded_dim(n)
                    st = self.__dict__.setdefault('_adopt', {})
                    buf = st.get('buf')
                    if buf is None or buf.shape[0] < M or tuple(buf.shape[1:]) != (Np, Np):
                        hit = self._factor_cache.get(0)
                        ok = hit is not None and hit[1].shape[0] >= M and tuple(hit[1].shape[1:]) == (Np, Np)
                        buf = hit[1] if ok else e.empty(M, Np, Np)
                        st.clear()
                        st['buf'] = buf
                        self._factor_cache.pop(0, None)   # (its buffer is being rewritten)
                    buf[k].copy_(Aslot)
                    st[k] = (k, nd.name, tuple(np.asarray(nd.length, float)), float(nd.nugget[0]), id(nd._input))
                    if all(j in st for j in range(M)):
                        self._factor_cache[0] = (tuple(st[j] for j in range(M)), buf)
                elif l == L - 1 and L == 2 and nd.prior_name != 'ref':
                    s_ = float(nd.scale[0])
                    ll = -0.5 * (n * float(np.log(s_)) + float(host[0]) + float(host[1]) / s_)
                    sig = (tuple(np.asarray(nd.length, float)), float(nd.nugget[0]), s_)
                    self.__dict__.setdefault('_adopt_ll', {})[k] = (ll, sig)
                return

    def _adopted_ll0(self):
        """Sum of the upper nodes' log-likelihoods handed over by the M-step, if it is still valid (same latents -- _attach
        drops it when they were re-uploaded -- and same hyper-parameters); used once."""
        got = self.__dict__.pop('_adopt_ll', None)
        upper = self.all_layer[1]
        if not got or len(got) != len(upper):
            return None
        tot = 0.0
        for k, nd in enumerate(upper):
            ll, sig = got[k]
            if sig != (tuple(np.asarray(nd.length, float)), float(nd.nugget[0]), float(nd.scale[0])):
                return None
            tot += ll
        return tot

    def stage_for_mstep(self, trusted=False):
        """Device views of every dense GP node's (input, global input, output) in the state the last sample() left --
        what kernel._stage() would upload from the numpy attributes _detach has just written, without the round trip
        through the host (24 small uploads per M-step at the bench shape).  {id(node): dict}; nodes with replicates,
        Vecchia nodes and likelihood nodes are left to their own staging.
        trusted: called by dgp.train's loop straight after sample(detach=False) -- the device state IS the truth (the numpy
        attributes have not even been refreshed yet), nothing is compared."""
        out = {}
        self.__dict__.pop('_adopt', None)
        self.__dict__.pop('_adopt_ll', None)
        L = len(self.all_layer)
        if trusted:
            if self.__dict__.get('F') is None:
                return out
            for l in range(L):
                for k, nd in enumerate(self.all_layer[l]):
                    if nd.type != 'gp' or nd.rep is not None:
                        continue
                    if l == 0:
                        Xl = self._x0[k]
                    else:
                        idx = np.asarray(nd.input_dim)
                        src = self.F[l - 1]
                        Xl = src if (len(idx) == src.shape[1] and np.array_equal(idx, np.arange(src.shape[1]))) \
                            else src[:, torch.as_tensor(idx, device=src.device)].contiguous()
                    if nd.vecch:
                        Xg = self._glob[(l, k)]
                        out[id(nd)] = dict(X=Xl if Xg is None else torch.cat((Xl, Xg), 1), y=self._node_y(l, k))
                    else:
                        out[id(nd)] = dict(Xl=Xl, Xg=self._glob[(l, k)], y=self._node_y(l, k), W=None)
            return out
        self.finish_detach()
        if not self.__dict__.get('_Fh') or self.__dict__.get('F') is None:   # (no sample() yet, or state dropped by pickling)
            return out
        Fh = self._Fh
        for l in range(L):
            for k, nd in enumerate(self.all_layer[l]):
                if nd.type != 'gp' or nd.rep is not None:
                    continue
                # (valid only while the numpy attributes still hold what _detach wrote / what was uploaded)
                const = self.__dict__.get('_const', {})
                y_ok = np.array_equal(np.asarray(nd.output, dtype=float).reshape(-1),
                                      Fh[l][:, k] if l < L - 1 else const.get(('y', k), (None,))[0])
                g_ok = nd._global_input is None or self._const_same(('g', l, k), nd._global_input, nd._private('global_input'))
                if l == 0:
                    x_ok = self._const_same(('x', k), nd._input, nd._private('input'))
                    Xl = self._x0[k]
                else:
                    idx = np.asarray(nd.input_dim)
                    mine = self.__dict__.get('_given_inputs', {}).get(id(nd))
                    x_ok = (mine is nd._input and nd._private('input')) or np.array_equal(nd._input, Fh[l - 1][:, idx])
                    src = self.F[l - 1]
                    Xl = src if (len(idx) == src.shape[1] and np.array_equal(idx, np.arange(src.shape[1]))) \
                        else src[:, torch.as_tensor(idx, device=src.device)].contiguous()
                if not (y_ok and g_ok and x_ok):
                    continue
                if nd.vecch:   # (kernel._vecch_stage orders these on the device)
                    Xg = self._glob[(l, k)]
                    out[id(nd)] = dict(X=Xl if Xg is None else torch.cat((Xl, Xg), 1), y=self._node_y(l, k))
                else:
                    out[id(nd)] = dict(Xl=Xl, Xg=self._glob[(l, k)], y=self._node_y(l, k), W=None)
        return out

    def _node_input(self, l, k, nd):
        """(Xloc tensor, colmap) of GP node k in layer l in the current state."""
        if l == 0:
            return self._x0[k], None
        return self.F[l - 1], np.asarray(nd.input_dim, dtype=np.int32)

    def _node_y(self, l, k):
        L = len(self.all_layer)
        return self._yy[k] if l == L - 1 else self.F[l][:, k].contiguous()

    # ------------------------------------------------------------------ sampling
    def sample(self, burnin=0, detach=True, trusted=False):
        """ESS-within-Gibbs over the layers (imputation.py:22-42).  detach=False (dgp.train's own loop): the numpy attributes of
        the nodes are refreshed later, by finish_detach() -- the M-step's first lock-step round is launched from the device
        state in between, so that the device works while the host copies 3 MB of latents back and rewrites the attributes
        (a 4-ms hole per iteration at n = 50 000).  Everything that reads the attributes calls finish_detach() first."""
        self.finish_detach()
        self._attach(trusted)
        n_layer = len(self.all_layer)
        if n_layer > 2 and self._sample_queued_deep(burnin + 1):   # every sweep of every hidden layer queued on the device
            self._detach(defer=not detach)
            return
        first, ahead = self._sample_queued(burnin + 1) if n_layer == 2 else (0, None)   # sweeps done without host round trips
        if first > burnin:
            self._detach(defer=not detach)
            return
        if ahead is None:
            ahead = self._prior_draws_ahead(burnin + 1) if n_layer > 1 else None
        else:
            ahead = ahead[first:]   # (the prior draws of every sweep were made for the queue)
        for sweep in range(burnin + 1 - first):
            for l in range(n_layer - 1):
                upper = self.all_layer[l + 1]
                hetero = any(nd.type == 'likelihood' and getattr(nd, 'exact_post_idx', None) is not None for nd in upper)
                if self.block and not hetero:   # (imputation.py:35-42: an exact-posterior likelihood forces node-wise updates)
                    self.one_sample_block(l, nu=ahead[sweep] if (ahead is not None and l == 0) else None)
                else:
                    for k in range(len(self.all_layer[l])):
                        self.one_sample(l, k)
        self._detach(defer=not detach)

    queued_calls = 0        # sample() calls that ran through the device queue (the tests read it)
    queue_max_batches = 2   # speculative batches queued per update (batch, then batch_next): 12 + 4 proposals; an update that
                            # needs more (a few per cent) is finished by the host loop and the rest of the I-step queued anew

    def _sample_queued(self, sweeps):
        """All `sweeps` block updates of a two-layer model's hidden layer queued on the device without a single host
        synchronisation (dgpamd_ess_queue; imputation.py:22-119): the accept / shrink loop runs in one-thread kernels on a
        device state, the uniform stream is uploaded ahead and consumed exactly as the sequential loop would.  Returns the
        number of sweeps completed and the prior draws of all sweeps (an update that ran out of queued batches or uniforms is finished by the host loop,
        and the remaining sweeps take the ordinary path).  (0, None): not applicable (Vecchia / likelihood / reference-prior
        nodes, node-wise updates)."""
        if not self.block or not getattr(self, 'queued', True):
            return 0, None
        layer, upper = self.all_layer[0], self.all_layer[1]
        if not self._queue_applies(0):
            return 0, None
        gp_top = all(nd.type == 'gp' for nd in upper)
        if self._ll_cache.get(0) is None and gp_top:   # the first threshold's log-likelihood: handed over by the M-step if it can be ...
            v = self._adopted_ll0()
            if v is not None:
                self._ll_cache[0] = v
        if self._ll_cache.get(0) is None and gp_top and not any(nd.vecch for nd in layer + upper):   # ... else factored together with the prior's matrices
            self._want_ll0 = list(enumerate(upper))
        nu = self._prior_draws_ahead(sweeps, prefetch=False)   # (sweeps, n, M); the next call's normals are started below
        self.__dict__.pop('_want_ll0', None)
        if nu is None:
            return 0, None
        e = self.engine
        F = self.F[0]
        n, M = F.shape
        plan = self._queue_plan(0)
        b0, bn, qmax = self._queue_batches(0)
        per = 2 + (b0 - 1) + (qmax - 1) * bn + 2
        scales = [float(nd.scale[0]) if nd.type == 'gp' else 1.0 for nd in upper]
        first = 0
        begun = False
        open_update = None   # the state of an update a queue left open (out of queued batches / of uploaded uniforms): the next queue goes on with it
        while first < sweeps:
            us = self.draws.uniform_peek((sweeps - first) * per)
            cur = self._ll_cache.get(0)
            if open_update is not None:
                plan.resume_state(open_update)
                plan.queue(F, nu[first:], scales, us, 0, None, 2, bn, qmax, fresh=False)
                open_update = None
            else:
                plan.queue(F, nu[first:], scales, us, 0, cur, cur is None, bn, qmax)
            if not begun:   # the next call's normals: generated by a background thread while this one waits in fetch()
                self.draws.prefetch(sweeps * M * n, engine=e)   # (started only now: it would fight the launches above for the interpreter)
                self.queued_calls += 1
                begun = True
            st = plan.fetch()   # the one synchronisation of the queue (of the whole I-step unless an update was left open)
            status, done = int(st['status']), int(st['updates'])
            self.draws.uniform_take(int(st['cursor']))
            self.stats['proposals'] += int(st['proposals'])
            self.stats['batches'] += int(st['batches'])
            self.stats['updates'] += done
            self._ll_cache[0] = float(st['ll'])
            first += done
            if status == 0:
                break
            if status == 2:
                raise_not_pd(int(st['info']))
            if status == 4:   # the uploaded uniforms ran out between two updates
                if not us:
                    raise RuntimeError('injected uniform stream exhausted')
                continue
            # the open update (out of queued batches, status 3, or of uploaded uniforms, status 1)
            if status == 1 and not us:
                raise RuntimeError('injected uniform stream exhausted')
            if os.environ.get('DGPAMD_ESS_RESUME', '1') != '0':
                # round 6: the next queue continues it on the device (dgpamd_ess_queue, compute_ll0 = 2) -- the host loop's finish cost an idle
                # millisecond 0.4 times per iteration at the bench shape (profiles/r06_idle_gaps.txt)
                open_update = st
                continue
            # (DGPAMD_ESS_RESUME=0, rounds 3-5: the host loop finishes it, the remaining sweeps are queued anew)
            self.one_sample_block(0, nu=nu[first], resume=dict(log_y=float(st['log_y']), theta=float(st['theta']), lo=float(st['lo']),
                                                                 hi=float(st['hi']), pending=bool(st['pending'])))
            first += 1
        return sweeps, nu

    def _queue_applies(self, l):
        """Can the updates of hidden layer l run through dgpamd_ess_queue?  Block updates; GP nodes in the layer; above it GP
        nodes without a reference prior (its constant depends on the proposal and is evaluated on the host) or likelihood
        nodes whose log-density the library evaluates (_lik_device_kind);
        the layer's own nodes all dense or all Vecchia.  With the Vecchia rows split over ranks (dist.split_training(rows=True))
        every rank queues the same launches on its own rows and the engine's reduce hook sums each batch's partial sums
        over the ranks on the stream (Engine.use_dist_reduce): the decisions are taken on identical numbers everywhere."""
        if not self.block or not getattr(self, 'queued', True):
            return False
        layer, upper = self.all_layer[l], self.all_layer[l + 1]
        if any(nd.type != 'gp' for nd in layer):
            return False
        for nd in upper:
            if nd.type == 'likelihood':
                if self._lik_device_kind(nd) is None:
                    return False
            elif nd.type != 'gp' or nd.prior_name == 'ref':
                return False
        if len({bool(nd.vecch) for nd in layer}) != 1 or len(layer) > 64:
            return False
        return True

    @staticmethod
    def _lik_device_kind(nd):
        """Name of the library's log-density for likelihood node nd (dgpamd_lik_loglik), or None: Hetero (its mean latent is
        drawn from an exact posterior, node-wise updates) and user plugins stay with the host protocol llik()."""
        from . import likelihood_class as lc
        if getattr(nd, 'exact_post_idx', None) is not None or os.environ.get('DGPAMD_LIK_HOST') == '1':   # (the switch: comparison runs)
            return None
        if type(nd) in (lc.Poisson, lc.NegBin, lc.ZIP, lc.ZINB):
            return nd.name
        if type(nd) is lc.Categorical:
            if nd.num_classes == 2:
                return 'logit' if nd.link == 'logit' else 'probit'
            return 'robustmax' if nd.link == 'robustmax' else 'softmax'
        return None

    def _lik_dev(self, nd):
        """Device-side description of likelihood node nd for Engine.lik_loglik / the queue: observations, replicate map."""
        hit = self.__dict__.setdefault('_lik_cache', {}).get(id(nd))
        if hit is None or hit[0] is not nd.output or hit[1] is not nd.rep:
            e = self.engine
            if getattr(nd, 'num_classes', 0):   # (the device log-density indexes the latent columns by the label)
                lab = np.asarray(nd.output, dtype=float).reshape(-1)
                if not (np.all(lab == np.floor(lab)) and lab.min(initial=0) >= 0 and lab.max(initial=0) < nd.num_classes):
                    raise ValueError('Categorical likelihood: class labels must be whole numbers in [0, %d)' % nd.num_classes)
            d = dict(kind=self._lik_device_kind(nd), y=e.tensor(np.asarray(nd.output, dtype=float).reshape(-1)),
                     rep=None if nd.rep is None else torch.as_tensor(np.asarray(nd.rep, dtype=np.int64), device=self.F[0].device),
                     classes=getattr(nd, 'num_classes', 0), par=getattr(nd, 'robustmax_eps', 0.0))
            hit = self._lik_cache[id(nd)] = (nd.output, nd.rep, d)
        return hit[2]

    def _queue_batches(self, l):
        """(first batch, later batches, batches queued per update) of the device queue for hidden layer l.  Dense nodes upstairs:
        the sampler's settings (12, then 4; two batches queued) -- a factorisation batch is latency-bound, so wide batches cost
        little.  Vecchia nodes upstairs: an evaluation is throughput-bound (0.1 ms of the whole device per candidate at
        n = 50 000) and a queued batch costs no host round trip, so narrow batches waste fewer candidates: 6, then 3, five
        queued (cfg4 I-step 50 -> 40 ms; profiles/r03_cfg4_vecchia.txt).  Explicit settings of the caller are kept."""
        upper = self.all_layer[l + 1]
        default = self._batch_default and self.batch == 10 and self.batch_next == 6 and self.queue_max_batches == 2
        env = os.environ.get('DGPAMD_ESS_BATCH')   # "first,next,queued": tuning runs (tools/gpu_ess_batch_choice.py)
        if default and env:
            b0, bn, qm = (int(v) for v in env.split(','))
            return b0, bn, qm
        if default and all(nd.type == 'gp' and nd.vecch for nd in upper):
            return 6, 3, 5
        # dense nodes of cfg3's size: a candidate's factorisation (n^3 / 3 at the engine's ~45 TFLOP/s: 0.9 ms at n = 5000) costs as
        # much as the whole pivot chain (n / 64 steps of 15 us), so wasted candidates are no longer free: 1.29 -> 1.41 SI it/s
        if default and all(nd.type == 'gp' for nd in upper) and self.F[l].shape[0] >= 4000:
            return 6, 3, 5
        return self.batch, int(self.batch_next) if self.batch_next else self.batch, self.queue_max_batches

    def _queue_plan(self, l):
        """The dgpamd_ess_queue arguments of hidden layer l (nodes of layer l+1 as the target's likelihood), rebuilt when
        their hyper-parameters, buffers, orderings or the batch size change.  Outputs of a hidden layer upstairs live in
        buffers of their own (`ybuf`), refreshed from the latents by _queue_refresh_y before every update."""
        e = self.engine
        upper = self.all_layer[l + 1]
        n, M = self.F[l].shape
        last = l + 1 == len(self.all_layer) - 1
        key = tuple(('lik', self._lik_dev(nd)['y'].data_ptr(), tuple(np.asarray(nd.input_dim).tolist())) if nd.type == 'likelihood' else
                    (tuple(np.asarray(nd.length, float)), float(nd.nugget[0]), self._node_y(l + 1, k).data_ptr() if last else None,
                     None if self._glob[(l + 1, k)] is None else self._glob[(l + 1, k)].data_ptr(),
                     (id(nd.ord), id(nd.NNarray), None if nd.rep is None else id(nd.W_diag)) if nd.vecch else None)
                    for k, nd in enumerate(upper)) + (self._queue_batches(l)[0], n, M, ddist.vecchia_rows(n) if ddist.rows_split() else None)
        hit = self._ess_plans.get(('queue', l))
        if hit is None or hit[0] != key:
            nodes, ybuf = [], {}
            for k, nd in enumerate(upper):
                if nd.type == 'likelihood':
                    nodes.append(dict(colmap=np.asarray(nd.input_dim, dtype=np.int32), lik=self._lik_dev(nd)))
                    continue
                y = self._node_y(l + 1, k) if last else e.empty(n)
                d = dict(kind=nd.name, colmap=np.asarray(nd.input_dim, dtype=np.int32), Xglob=self._glob[(l + 1, k)], length=nd.length,
                         nugget=nd.nugget[0], W=None if nd.rep is None else e.tensor(nd.W_diag), y=y)
                if nd.vecch:
                    od = nd.ord_dev()
                    d['vecch'] = dict(ord=od, nn=nd.nn_dev(), nd=e.tensor(np.ones(n) if nd.rep is None else nd.W_diag),
                                      y=y[od].contiguous() if last else e.empty(n),
                                      rows=ddist.vecchia_rows(n) if ddist.rows_split() else None)
                if not last:
                    ybuf[k] = (y, d['vecch']['y'] if nd.vecch else None, d['vecch']['ord'] if nd.vecch else None)
                nodes.append(d)
            plan = e.ess_queue_plan(n, M, nodes, self._queue_batches(l)[0])
            plan.ybuf = ybuf
            self._ess_plans[('queue', l)] = hit = (key, plan)
        return hit[1]

    def _queue_refresh_y(self, l, plan):
        """The outputs of the nodes above hidden layer l are latents themselves when layer l+1 is hidden: copy the current
        columns into the buffers the queued launches read (device copies on the engine's stream)."""
        for k, (y, yord, od) in plan.ybuf.items():
            col = self.F[l + 1][:, k]
            y.copy_(col)
            if yord is not None:
                yord.copy_(col[od])

    def _prior_draw_queued(self, l, Z, plan):
        """_prior_draw(l) for a queue: nu (n, M) from the normals Z (M, n; device) with no host synchronisation -- a failed
        factorisation is noted in the queue's device state (status 2) instead of raised here."""
        e = self.engine
        layer = self.all_layer[l]
        n, M = self.F[l].shape
        if layer[0].vecch:
            return self._vecchia_draws(l, list(range(M)), Z[:, None, :].contiguous())[:, 0].t().contiguous()
        Np = e.padded_dim(n)
        buf = e.workspace(('qprior', l, n, M), M * Np * Np * 8)[:M * Np * Np * 8].view(torch.float64).view(M, Np, Np)
        for k, nd in enumerate(layer):
            Xl, cm = self._node_input(l, k, nd)
            e.kmatrix(nd.name, Xl, cm, self._glob[(l, k)], nd.length, nd.nugget[0], out=buf[k], full=False)
        _, info = e.potrf(n, buf, batch=M)
        plan.note_info(info)
        out = e.trmv_lower(n, buf, [float(nd.scale[0]) for nd in layer], Z, batch=M)
        return out.t().contiguous()

    def _sample_queued_deep(self, sweeps):
        """sample() for a model with more than one hidden layer with every update of every sweep queued on the device
        (imputation.py:22-119): per sweep and hidden layer -- the layer's prior draw from its CURRENT inputs (layers above
        the first: K assembly, factorisation and triangular product queued as well), the refresh of the output buffers of the
        nodes upstairs, ONE update through dgpamd_ess_queue continuing the shared device state -- and one fetch at the end.
        The normals are taken from the stream in the order the host loop takes them.  An update left open (out of queued
        batches or uniforms) is finished by the host loop and the rest is queued anew.  Returns False when a layer does not
        qualify (the host loop then runs the whole call)."""
        L = len(self.all_layer)
        hidden = list(range(L - 1))
        if not all(self._queue_applies(l) for l in hidden):
            return False
        e = self.engine
        n = self.F[0].shape[0]
        injected = self.draws._z is not None
        # normals, in the host loop's order: layer 0 of every sweep first (one upload; _prior_draws_ahead) unless the stream is
        # an injected one, then sweep by sweep the deeper layers
        nu0 = None if injected else self._prior_draws_ahead(sweeps, prefetch=False)
        Zs = {}
        for s_ in range(sweeps):
            for l in hidden:
                if l == 0 and nu0 is not None:
                    continue
                Zs[(s_, l)] = np.stack([self.draws.normal(n) for _ in self.all_layer[l]])
        Zdev = {key: e.tensor(z) for key, z in Zs.items()}
        ops = [(s_, l) for s_ in range(sweeps) for l in hidden]
        qb = {l: self._queue_batches(l) for l in hidden}
        per = max(2 + (b0 - 1) + (qmax - 1) * bn + 2 for b0, bn, qmax in qb.values())
        plans = {l: self._queue_plan(l) for l in hidden}
        pos = 0
        while pos < len(ops):
            us = self.draws.uniform_peek((len(ops) - pos) * per)
            lead = plans[ops[pos][1]]
            nus = {}
            # The window's device state is reset BEFORE its first operation: that operation's prior factorisation (a deeper
            # layer's, or layer 0's under an injected stream) notes its info word in the state, and a reset behind it would
            # drop a non-positive-definite prior silently (the queue would carry on with nu from a failed factor instead of
            # raising LinAlgError like the host loop and the reference, imputation.py:54-63).
            lead.reset_state(0, None)
            for j in range(pos, len(ops)):
                s_, l = ops[j]
                plan = plans[l]
                if plan is not lead:
                    lead.upload_uniforms(us)
                    plan.share_with(lead)
                if l == 0 and nu0 is not None:
                    nu = nu0[s_]
                else:
                    nu = self._prior_draw_queued(l, Zdev[(s_, l)], lead)
                nus[j] = nu
                self._queue_refresh_y(l, plan)
                scales = [float(nd.scale[0]) if nd.type == 'gp' else 1.0 for nd in self.all_layer[l + 1]]
                plan.queue(self.F[l], nu[None], scales, us, 0, None, True, qb[l][1], qb[l][2], fresh=False)
            st = lead.fetch()
            status, done = int(st['status']), int(st['updates'])
            self.draws.uniform_take(int(st['cursor']))
            self.stats['proposals'] += int(st['proposals'])
            self.stats['batches'] += int(st['batches'])
            self.stats['updates'] += done
            self._ll_cache = {}
            pos += done
            if status == 0:
                break
            if status == 2:
                raise_not_pd(int(st['info']))
            if status == 4:
                if not us:
                    raise RuntimeError('injected uniform stream exhausted')
                continue
            s_, l = ops[pos]   # the open update: finished by the host loop from the device's bracket
            self.one_sample_block(l, nu=nus[pos], resume=dict(log_y=float(st['log_y']), theta=float(st['theta']), lo=float(st['lo']),
                                                                hi=float(st['hi']), pending=bool(st['pending'])))
            pos += 1
        if nu0 is not None:
            self.draws.prefetch(sweeps * self.F[0].shape[1] * n, engine=e)
        self.queued_calls += 1
        return True

    def _layer_factors(self, l, dense):
        """Cholesky factors of the dense nodes `dense` of layer l in ONE batched buffer (stride Np^2) so that the draws
        are a single batched triangular product; the buffer is reused while inputs and hyper-parameters are unchanged
        (first layer: the whole I-step and beyond; deeper layers: their inputs change every sweep).
        If self._want_ll0 lists the GP nodes above layer 0 whose log-likelihood of the CURRENT latents the sampler needs
        next (the first slice threshold of an I-step, imputation.py:70-78), their matrices ride along as extra members of the
        same batched factorisation -- one launch instead of two chain-bound ones -- and the sum lands in self._ll_cache[0]."""
        e = self.engine
        layer = self.all_layer[l]
        n = self.F[l].shape[0]
        Np = e.padded_dim(n)
        sigs = tuple((k, layer[k].name, tuple(np.asarray(layer[k].length, float)), float(layer[k].nugget[0]),
                      id(layer[k]._input) if l == 0 else None) for k in dense)   # (an input edited in place empties this cache: _attach)
        hit = self._factor_cache.get(l)
        if l != 0 or hit is None or hit[0] != sigs:
            extra = self.__dict__.pop('_want_ll0', None) if l == 0 else None
            extra = extra if extra and len(dense) + len(extra) <= 64 and all(len(nd.output) == n for _, nd in extra) else []
            nb = len(dense) + len(extra)
            buf = hit[1] if hit is not None and hit[1].shape[0] == nb else e.empty(nb, Np, Np)
            for j, k in enumerate(dense):
                nd = layer[k]
                Xl, cm = self._node_input(l, k, nd)
                e.kmatrix(nd.name, Xl, cm, self._glob[(l, k)], nd.length, nd.nugget[0], out=buf[j], full=False)
            for j, (k, nd) in enumerate(extra):   # K of an upper node at the current latents, its output riding along
                Xl, cm = self._node_input(1, k, nd)
                e.kmatrix(nd.name, Xl, cm, self._glob[(1, k)], nd.length, nd.nugget[0],
                          W=None if nd.rep is None else e.tensor(nd.W_diag), out=buf[len(dense) + j], full=False, Y=self._node_y(1, k))
            lls = []
            for c0 in range(0, nb, 64):
                n1 = min(64, nb - c0)
                logdet, info = e.potrf(n, buf[c0:c0 + n1], batch=n1)
                for j, (k, nd) in enumerate(extra):
                    lls.append(e.loglik_finish(n, buf[len(dense) + j], logdet[len(dense) + j:len(dense) + j + 1], nd.scale[0]))
                got = e.fetch(torch.cat([info.to(torch.float64)] + lls))   # one synchronisation for both
                if got[:n1].any():
                    bad = got[:n1]
                    raise_not_pd(int(bad[bad != 0][0]))
                if extra:
                    self._ll_cache[0] = float(got[n1:].sum())
            self._factor_cache[l] = (sigs, buf)
        return self._factor_cache[l][1]

    def _prior_draws_ahead(self, sweeps, prefetch=True):
        """Layer 0's prior draws for all `sweeps` sweeps of a sample() call at once: its factors depend only on X and
        the hyper-parameters, so the normals of every sweep are uploaded in one copy and the triangular products are
        queued back to back.  Consumes the normal stream in the order the sequential loop would as long as layer 0 is
        the only hidden layer; with an injected stream and deeper hierarchies it is therefore not used."""
        layer = self.all_layer[0]
        if not self.block or any(nd.type != 'gp' for nd in layer):
            return None
        all_vecch = all(nd.vecch for nd in layer)
        if any(nd.vecch for nd in layer) and not all_vecch:
            return None
        if any(nd.type == 'likelihood' and getattr(nd, 'exact_post_idx', None) is not None for nd in self.all_layer[1]):
            return None   # node-wise updates (sample())
        if self.draws._z is not None and len(self.all_layer) > 2:
            return None
        e = self.engine
        n, M = self.F[0].shape
        if M > 64:
            return None
        if all_vecch:
            # Vecchia: M x sweeps independent sparse forward substitutions, all in one launch
            if self.draws._z is None:
                Zd = self.draws.normals_device(e, sweeps * M * n).view(sweeps, M, n)
            else:
                Zd = e.tensor(np.stack([np.stack([self.draws.normal(n) for _ in range(M)]) for _ in range(sweeps)]))
            xs = self._vecchia_draws(0, list(range(M)), Zd.permute(1, 0, 2).contiguous())   # (M, sweeps, n)
            if prefetch:
                self.draws.prefetch(sweeps * M * n, engine=e)
            return xs.permute(1, 2, 0).contiguous()
        buf = self._layer_factors(0, list(range(M)))
        if self.draws._z is None:
            Zd = self.draws.normals_device(e, sweeps * M * n).view(sweeps, M, n)
        else:
            Zd = e.tensor(np.stack([np.stack([self.draws.normal(n) for _ in range(M)]) for _ in range(sweeps)]))
        scales = [float(nd.scale[0]) for nd in layer]
        out = e.empty(sweeps, M, n)
        for s_ in range(sweeps):
            e.trmv_lower(n, buf, scales, Zd[s_], batch=M, out=out[s_])
        if prefetch:
            self.draws.prefetch(sweeps * M * n, engine=e)   # the next call's normals, generated while the device works
        return out.transpose(1, 2).contiguous()

    def _prior_draw(self, l, cols=None):
        """nu[:, k] = chol(scale_k K_k) z_k for the nodes `cols` of layer l (default: all)
        (imputation.py:54-63,166-172, functions.py:113-121)."""
        e = self.engine
        layer = self.all_layer[l]
        n, M = self.F[l].shape
        Np = e.padded_dim(n)
        nu = e.zeros(n, M)
        cols = list(range(M)) if cols is None else list(cols)
        dense = [k for k in cols if not layer[k].vecch]
        Z = {k: self.draws.normal(n) for k in cols}
        if dense:
            buf = self._layer_factors(l, dense)
            Zd = e.tensor(np.stack([Z[k] for k in dense]))
            scales = [float(layer[k].scale[0]) for k in dense]
            for c0 in range(0, len(dense), 64):
                nb = min(64, len(dense) - c0)
                out = e.trmv_lower(n, buf[c0:c0 + nb], scales[c0:c0 + nb], Zd[c0:c0 + nb], batch=nb)
                nu[:, torch.as_tensor(dense[c0:c0 + nb], device=nu.device)] = out.t()
        vec = [k for k in cols if layer[k].vecch]
        if vec:
            xs = self._vecchia_draws(l, vec, e.tensor(np.stack([Z[k] for k in vec])[:, None, :]))   # (nodes, 1, n)
            for j, k in enumerate(vec):
                nu[:, k] = xs[j, 0]
        return nu

    def _vecchia_draws(self, l, nodes, Zd):
        """fmvn_sp (vecchia.py:133-140) for the Vecchia nodes `nodes` of layer l and Zd.shape[1] normal vectors each
        (Zd: (nodes, draws, n) on the device): the rows of the sparse inverse factors (one launch per node), then ALL
        forward substitutions -- independent serial chains -- side by side in one launch.  Returns (nodes, draws, n) in
        the original point order."""
        e = self.engine
        layer = self.all_layer[l]
        if len({layer[k].NNarray.shape[1] for k in nodes}) != 1:     # different conditioning sizes: one by one
            return torch.stack([self._vecchia_draws(l, [k], Zd[j:j + 1])[0] for j, k in enumerate(nodes)])
        Lms, NNs, sc, rev = [], [], [], []
        for k in nodes:
            nd = layer[k]
            Xl, cm = self._node_input(l, k, nd)
            X = Xl if cm is None else Xl[:, torch.as_tensor(cm, device=Xl.device, dtype=torch.long)]
            if self._glob[(l, k)] is not None:
                X = torch.cat((X, self._glob[(l, k)]), 1)
            od = nd.ord_dev()
            NN = nd.nn_dev()
            Lms.append(e.vecchia_lmatrix(nd.name, X[od].contiguous(), NN, nd.length, nd.nugget[0]))
            NNs.append(NN)
            sc.append(1.0 / np.sqrt(nd.scale[0]))
            rev.append(nd.rev_ord_dev())
        NNall = torch.stack(NNs)
        # the substitutions run level-scheduled; the schedule depends on the neighbour arrays only (rebuilt when they are)
        sig = tuple(id(layer[k].NNarray) for k in nodes)
        hit = self.__dict__.setdefault('_sp_levels', {}).get(l)
        if hit is None or hit[0] != sig:
            hit = self._sp_levels[l] = (sig, e.vecchia_levels(NNall))
        xs = e.vecchia_spsolve_levels(torch.stack(Lms), NNall, sc, Zd.contiguous(), hit[1])
        return torch.stack([xs[j][:, rev[j]] for j in range(len(nodes))])

    def _upper_loglik(self, l, FP, only=None):
        """sum over the nodes of layer l+1 (or those listed in `only`) of their log-likelihood for each candidate
        block FP[b] (imputation.py:70-78,91-106).  Returns (ll (B,) numpy, info (B,) numpy)."""
        e = self.engine
        B = FP.shape[0]
        upper = self.all_layer[l + 1]
        dev_terms, infos, vec_terms = [], [], []
        host = np.zeros(B)
        FPh = None
        for k, nd in enumerate(upper):
            if only is not None and k not in only:
                continue
            if nd.type == 'gp' and not nd.vecch:
                ll, info = e.loglik(nd.name, FP, np.asarray(nd.input_dim, dtype=np.int32), self._glob[(l + 1, k)], nd.length,
                                    nd.nugget[0], nd.scale[0], self._node_y(l + 1, k),
                                    W=None if nd.rep is None else e.tensor(nd.W_diag), batch=B,
                                    A=e.workspace(('essA', FP.shape[1], self.batch), self.batch * e.padded_dim(FP.shape[1]) ** 2 * 8),
                                    ll=e.empty(B), info=e.empty(B, dtype=torch.int32))
                dev_terms.append(ll)
                infos.append(info)
                if nd.prior_name == 'ref':
                    if FPh is None:
                        FPh = FP.cpu().numpy()
                    host += self._ref_prior_terms(nd, FPh)
            elif nd.type == 'gp':
                # (index arrays, the 10-MB neighbour array and the nugget weights stay on the device while the node keeps them)
                sig = (id(nd.ord), id(nd.NNarray), id(nd.input_dim), None if nd.rep is None else id(nd.W_diag), FP.shape[1])
                hit = self.__dict__.setdefault('_vecch_dev', {}).get((l + 1, k))
                if hit is None or hit[0] != sig:
                    hit = (sig, (torch.as_tensor(np.asarray(nd.input_dim), device=FP.device, dtype=torch.long),
                                 nd.ord_dev(), nd.nn_dev(),
                                 e.tensor(np.ones(FP.shape[1]) if nd.rep is None else nd.W_diag)))
                    self._vecch_dev[(l + 1, k)] = hit
                cm, od, NN, nd_diag = hit[1]
                ysrc = self._node_y(l + 1, k)
                if l + 1 == len(self.all_layer) - 1:   # (observed outputs: ordered once per ordering, not once per batch)
                    yhit = self.__dict__.setdefault('_vecch_y', {}).get((l + 1, k))
                    if yhit is None or yhit[0] is not ysrc or yhit[1] is not od:
                        yhit = self._vecch_y[(l + 1, k)] = (ysrc, od, ysrc[od].contiguous())
                    y = yhit[2]
                else:
                    y = ysrc[od].contiguous()
                # the ordered inputs of ALL candidates in at most three device operations (was five small ones per candidate:
                # the host could not keep the device busy between the row kernels); one when the node takes every column in order
                ident = FP.shape[2] == len(nd.input_dim) and np.array_equal(np.asarray(nd.input_dim), np.arange(FP.shape[2]))
                Xall = FP if ident else FP[:, :, cm]
                if self._glob[(l + 1, k)] is not None:
                    Xall = torch.cat((Xall, self._glob[(l + 1, k)].unsqueeze(0).expand(B, -1, -1)), 2)
                Xall = Xall[:, od].contiguous()
                lo, hi = ddist.vecchia_rows(NN.shape[0])   # (all rows unless dist.split_training(rows=True) on several ranks)
                o = e.vecchia_llik_batch(nd.name, Xall, y, NN[lo:hi], nd.length, nd.nugget[0], nd_diag)
                if ddist.rows_split():   # one all-reduce of the batch's 2 B sums: every rank takes the same accept decisions
                    o = ddist.allreduce_sum_vector(o.reshape(-1)).reshape(B, 2)
                vec_terms.append((o, float(nd.scale[0])))   # (-0.5 (logdet + quad / scale): closed on the host below)
                if nd.prior_name == 'ref':
                    if FPh is None:
                        FPh = FP.cpu().numpy()
                    host += self._ref_prior_terms(nd, FPh)
            elif self._lik_device_kind(nd) is not None:
                # likelihood node the library knows (likelihood_class.py llik()): summed on the device, in the order the queue uses
                dev_terms.append(e.lik_loglik(self._lik_dev(nd), np.asarray(nd.input_dim, dtype=np.int32), FP))
            else:
                # likelihood node: host plugin protocol llik() on .input (likelihood_class.py:30-90)
                if FPh is None:
                    FPh = FP.cpu().numpy()
                for b in range(B):
                    nd.input = FPh[b][nd.rep, :][:, nd.input_dim] if nd.rep is not None else FPh[b][:, nd.input_dim]
                    host[b] += float(np.sum(nd.llik()))
        if vec_terms:
            # the Vecchia nodes' (quad, logdet) sums: ONE fetch, combined here -- a torch expression on the device would be
            # three more launches per batch (and 0.15 s of lazy code loading the first time a process runs it)
            sums = e.fetch(vec_terms[0][0] if len(vec_terms) == 1 else torch.cat([o.reshape(-1) for o, _ in vec_terms])).reshape(-1, B, 2)
            for (_, sc_), o in zip(vec_terms, sums):
                host = host + -0.5 * (o[:, 1] + o[:, 0] / sc_)
        if dev_terms:
            # results come back through the library's pinned buffer (one sync); a single GP node upstairs -- the usual
            # case -- needs no device-side packing at all
            if len(dev_terms) == 1 and len(infos) == 1:
                ll_h, info_h = e.fetch_ll_info(dev_terms[0], infos[0])
                host, info = host + ll_h, info_h.astype(np.float64)
            else:
                host = host + e.fetch(dev_terms[0] if len(dev_terms) == 1 else torch.stack(dev_terms).sum(0))
                if infos:
                    info = e.fetch(infos[0]).astype(np.float64) if len(infos) == 1 else \
                        e.fetch(torch.stack(infos).amax(0)).astype(np.float64)
                else:
                    info = np.zeros(B)
        else:
            info = np.zeros(B)
        return host, info

    @staticmethod
    def _ref_prior_terms(nd, FPh):
        """Reference-prior term of the ESS target for every candidate block (kernel_class.py:489-491,507-509): the prior's
        scaling constant depends on the node's inputs, i.e. on the proposal."""
        keep, keep_cl = peek(nd, 'input'), getattr(nd, 'cl', None)
        out = np.empty(FPh.shape[0])
        for b in range(FPh.shape[0]):
            bind_private(nd, 'input', FPh[b][:, nd.input_dim])
            nd.compute_cl()
            out[b] = float(np.sum(nd.log_prior()))
        bind_private(nd, 'input', keep)
        nd.cl = keep_cl
        return out

    def _ess_plan(self, l):
        """dgpamd_ess_update arguments for layer l when the layer above is ONE dense GP node without a reference prior
        (else None: the general loop below handles several / Vecchia / likelihood nodes).  Rebuilt when the node's
        hyper-parameters, the batch size or the device state change."""
        upper = self.all_layer[l + 1]
        if len(upper) != 1 or l + 1 != len(self.all_layer) - 1:   # (a hidden layer's output is itself being sampled)
            return None
        nd = upper[0]
        if nd.type != 'gp' or nd.vecch or nd.prior_name == 'ref':
            return None
        n, M = self.F[l].shape
        y = self._node_y(l + 1, 0)
        key = (tuple(np.asarray(nd.length, float)), float(nd.nugget[0]), self.batch, y.data_ptr(),
               None if self._glob[(l + 1, 0)] is None else self._glob[(l + 1, 0)].data_ptr())
        hit = self._ess_plans.get(l)
        if hit is None or hit[0] != key:
            W = None if nd.rep is None else self.engine.tensor(nd.W_diag)
            plan = self.engine.ess_plan(n, M, nd.name, np.asarray(nd.input_dim, dtype=np.int32), self._glob[(l + 1, 0)],
                                        nd.length, nd.nugget[0], W, y, self.batch)
            self._ess_plans[l] = hit = (key, plan)
        return hit[1]

    def one_sample_block(self, l, nu=None, resume=None):
        """Layer-wise ESS update of layer l given layer l+1 (imputation.py:44-119); nu: prior draw made ahead; resume:
        threshold, angle and bracket of an update that dgpamd_ess_queue left open."""
        e = self.engine
        F = self.F[l]
        if nu is None:
            nu = self._prior_draw(l)
        pending = False
        if resume is None:
            cur = self._ll_cache.get(l)
            if cur is None:
                ll, info = self._upper_loglik(l, F[None])
                if info[0] != 0:
                    raise_not_pd(int(info[0]))
                cur = ll[0]
            log_y = cur + np.log(self.draws.uniform_take(1)[0])
            theta = TWO_PI * self.draws.uniform_take(1)[0]
            lo, hi = theta - TWO_PI, theta
        else:
            log_y, theta, lo, hi, pending = resume['log_y'], resume['theta'], resume['lo'], resume['hi'], resume['pending']
        B = self.batch
        if resume is not None and self.batch_next:   # (the queue's batches were the wide ones: carry on as the host loop would)
            B = min(self.batch, int(self.batch_next))
        self.stats['updates'] += 1
        plan = self._ess_plan(l)
        if plan is not None:   # one dense GP node upstairs: the whole shrinking-bracket loop is one library call
            nd = self.all_layer[l + 1][0]
            while True:
                us = self.draws.uniform_peek(64)
                status, used, props, nbat, ll_acc, info, theta, lo, hi, pending = plan.run(
                    F, nu, nd.scale[0], log_y, theta, lo, hi, pending, us, self.batch_next)
                self.draws.uniform_take(used)
                self.stats['proposals'] += props
                self.stats['batches'] += nbat
                if status == 2:
                    raise_not_pd(info)
                if status == 0:
                    self._ll_cache[l] = ll_acc
                    self._ll_cache.pop(l - 1, None)
                    return
                if len(us) == used and self.draws._injected_u:
                    raise RuntimeError('injected uniform stream exhausted')
        if pending:   # (resumed after a fully rejected batch: its closing shrink is still due)
            theta, lo, hi = shrink(theta, lo, hi, self.draws.uniform_take(1)[0])
        while True:
            us = self.draws.uniform_peek(B - 1)
            thetas, brackets = speculative_angles(theta, lo, hi, us)
            nb = len(thetas)
            if self.batch_next:
                B = min(self.batch, int(self.batch_next))   # after a rejected batch the bracket is narrow
            FP = e.ess_propose(F, nu, thetas)
            ll, info = self._upper_loglik(l, FP)
            self.stats['batches'] += 1
            for b in range(nb):
                if info[b] != 0:
                    raise_not_pd(int(info[b]))
                if ll[b] > log_y:
                    self.draws.uniform_take(b)
                    self.stats['proposals'] += b + 1
                    F.copy_(FP[b])
                    self._ll_cache[l] = ll[b]
                    self._ll_cache.pop(l - 1, None)   # outputs of layer l feed layer l-1's upper log-likelihood
                    return
            self.draws.uniform_take(nb - 1)
            self.stats['proposals'] += nb
            theta, (lo, hi) = thetas[-1], brackets[-1]
            theta, lo, hi = shrink(theta, lo, hi, self.draws.uniform_take(1)[0])

    def _exact_posterior(self, l, k, lik):
        """Latent column k of layer l drawn from its exact conditional posterior under the likelihood node `lik`
        (imputation.py:141-164 -> Hetero.posterior, likelihood_class.py:134-243; dense mode)."""
        e = self.engine
        nd = self.all_layer[l][k]
        F = self.F[l]
        n = F.shape[0]
        if nd.vecch:
            # Vecchia form (imputation.py:143-160, likelihood_class.py:153-182): sparse factor of the stacked
            # [observations ; latents] vector, rows built and solved on the device in ordered coordinates
            import torch
            if getattr(nd, 'imp_NNarray', None) is None:
                nd.ord_nn(ord=nd.ord, NNarray=nd.NNarray, pointer=True)
            Fh = F.cpu().numpy()
            lik.input = Fh[lik.rep, :][:, lik.input_dim] if lik.rep is not None else Fh[:, lik.input_dim]
            yv = np.asarray(lik.output, float).reshape(-1)
            if lik.rep is not None:
                invg = 1.0 / np.exp(lik.input[:, 1])
                gam = 1.0 / np.bincount(lik.rep, weights=invg, minlength=n)
                yeff = np.bincount(lik.rep, weights=invg * yv, minlength=n) * gam
            else:
                gam, yeff = np.exp(lik.input[:, 1]), yv
            Xl, cm = self._node_input(l, k, nd)
            Xc = Xl if cm is None else Xl[:, torch.as_tensor(cm, device=Xl.device, dtype=torch.long)]
            if self._glob[(l, k)] is not None:
                Xc = torch.cat((Xc, self._glob[(l, k)]), 1)
            Xo = Xc[torch.as_tensor(nd.ord, device=Xc.device, dtype=torch.long)].contiguous()
            z = self.draws.normal(n)
            f = e.vecchia_post_het(nd.name, Xo, e.tensor(nd.imp_NNarray, dtype=torch.int64), nd.scale[0], nd.length,
                                   e.tensor(gam[nd.ord]), e.tensor(yeff[nd.ord]), e.tensor(z))
            F[:, k] = f[e.tensor(nd.rev_ord, dtype=torch.int64)]
            self._ll_cache.pop(l, None)
            self._ll_cache.pop(l - 1, None)
            return
        Xl, cm = self._node_input(l, k, nd)
        K = e.kmatrix(nd.name, Xl, cm, self._glob[(l, k)], nd.length, nd.nugget[0])   # full n x n, device
        Fh = F.cpu().numpy()
        lik.input = Fh[lik.rep, :][:, lik.input_dim] if lik.rep is not None else Fh[:, lik.input_dim]
        g, y = lik.posterior_terms(n)
        sd = self.draws.normal(2 * n).reshape(n, 2)
        f = e.post_het(K, nd.scale[0], e.tensor(g), e.tensor(y), e.tensor(sd))
        F[:, k] = f
        self._ll_cache.pop(l, None)
        self._ll_cache.pop(l - 1, None)

    def one_sample(self, l, k):
        """Node-wise ESS update of latent column k of layer l given the nodes of layer l+1 it feeds
        (imputation.py:121-221, the non-Hetero branch): same speculative batches, one column rotated."""
        e = self.engine
        F = self.F[l]
        linked = [j for j, nd in enumerate(self.all_layer[l + 1]) if k in np.asarray(nd.input_dim)]
        if len(linked) == 1:
            lik = self.all_layer[l + 1][linked[0]]
            if lik.type == 'likelihood' and getattr(lik, 'exact_post_idx', None) is not None:
                idx = int(np.where(np.asarray(lik.input_dim) == k)[0][0])
                if idx in np.asarray(lik.exact_post_idx):
                    self._exact_posterior(l, k, lik)
                    return
        nu = self._prior_draw(l, [k])
        ll, info = self._upper_loglik(l, F[None], only=linked)
        if info[0] != 0:
            raise_not_pd(int(info[0]))
        log_y = ll[0] + np.log(self.draws.uniform_take(1)[0])
        theta = TWO_PI * self.draws.uniform_take(1)[0]
        lo, hi = theta - TWO_PI, theta
        fk, nk = F[:, k:k + 1].contiguous(), nu[:, k:k + 1].contiguous()
        self.stats['updates'] += 1
        B = self.batch
        while True:
            us = self.draws.uniform_peek(B - 1)
            thetas, brackets = speculative_angles(theta, lo, hi, us)
            nb = len(thetas)
            if self.batch_next:
                B = min(self.batch, int(self.batch_next))
            col = e.ess_propose(fk, nk, thetas)            # (nb, n, 1)
            FP = F.unsqueeze(0).repeat(nb, 1, 1)
            FP[:, :, k] = col[:, :, 0]
            ll, info = self._upper_loglik(l, FP, only=linked)
            self.stats['batches'] += 1
            for b in range(nb):
                if info[b] != 0:
                    raise_not_pd(int(info[b]))
                if ll[b] > log_y:
                    self.draws.uniform_take(b)
                    self.stats['proposals'] += b + 1
                    F.copy_(FP[b])
                    self._ll_cache.pop(l, None)
                    self._ll_cache.pop(l - 1, None)
                    return
            self.draws.uniform_take(nb - 1)
            self.stats['proposals'] += nb
            theta, (lo, hi) = thetas[-1], brackets[-1]
            theta, lo, hi = shrink(theta, lo, hi, self.draws.uniform_take(1)[0])

    # ------------------------------------------------------------------ bookkeeping
    def key_stats(self):
        """Prediction statistics of every GP node (imputation.py:223-231)."""
        for layer in self.all_layer:
            for nd in layer:
                if nd.type == 'gp':
                    nd.compute_stats()

    def update_ord_nn(self):
        """Refresh ordering and neighbours, sharing them between sibling nodes that see identical
        scaled inputs (imputation.py:233-262)."""
        self.finish_detach()
        # the searches first, one behind the other on the device (their orderings drawn in node order, as the sequential loop
        # draws them), then the nodes that borrow a sibling's result
        waiting, borrowers = [], []
        for layer in self.all_layer:
            for k, nd in enumerate(layer):
                if nd.type != 'gp':
                    continue
                donor = None
                for j in range(k):
                    o = layer[j]
                    if o.type != 'gp' or not np.array_equal(nd.input_dim, o.input_dim) or not np.array_equal(nd.connect, o.connect):
                        continue
                    if len(nd.length) == 1 and len(o.length) == 1:
                        donor = (o, False)
                        break
                    if len(nd.length) != 1 and np.array_equal(nd.length, o.length):
                        donor = (o, True)
                        break
                if donor is None:
                    if len(waiting) == self.engine.MAILBOXES:   # (every mailbox holds a search: take the oldest one in)
                        waiting.pop(0)[1]()
                    slot = next(i for i in range(self.engine.MAILBOXES) if all(i != s_ for s_, _ in waiting))
                    fin = nd.ord_nn(slot=slot)
                    if fin is not None:
                        waiting.append((slot, fin))
                else:
                    borrowers.append((nd,) + donor)
        failed = None
        for _, fin in waiting:
            try:
                fin()
            except Exception as exc:   # noqa: BLE001  (take the other results in all the same: no mailbox stays occupied)
                failed = failed or exc
        if failed is not None:
            raise failed
        for nd, o, copy in borrowers:   # (a donor is an earlier node of the same layer: it has its arrays by now)
            if copy:
                nd.ord_nn(ord=o.ord.copy(), NNarray=o.NNarray.copy(), rev_ord=o.rev_ord.copy())
            else:
                nd.ord_nn(ord=o.ord, NNarray=o.NNarray, rev_ord=o.rev_ord)
