#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): SI iterations/sec (train) + predict pts/sec,
2-layer DGP n=2000 d=5, Matern-2.5 (configs[1]) on synthetic data.

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE stochastic-imputation iteration: I-step (ESS-within-Gibbs, ess_burn=10 -> 11
layer sweeps) + M-step (L-BFGS-B on every GP node).  All inputs are resident in HBM before
the timed region.  N > 1 (torch.distributed.run, one rank per GPU, RCCL): the SI chain of ONE
model does not shard, so every rank trains its own replica (weak scaling, no data-path
collective); the prediction leg shards the imputations over the ranks with one all-reduce.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_PEAK_TFLOPS = 78.6   # MI355X f64 matrix (= vector) peak, AMD datasheet; MI355X_MICROARCH.md has no f64 row
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def pmc_note(kernel, fallback):
    """Counter evidence for a kernel (profiles/r06_pmc_kernels.json, else round 5's; written by tools/pmc_summary.py from
    rocprofv3 --pmc passes of this build), as a sentence for the result line; `fallback` while that file has no entry."""
    for name in ('r06_pmc_kernels.json', 'r05_pmc_kernels.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                e = json.load(f).get(kernel)
            if e and e.get('note'):
                return e['note'] + ' (profiles/%s)' % name
        except (OSError, ValueError):
            pass
    return fallback


def synthetic(n, d, seed=2026):
    """X ~ U[0,1]^(n x d), standardised smooth non-stationary response."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(n, d))
    f = np.sin(1.0 / ((0.7 * X[:, 0] + 0.3) * (0.7 * X[:, 1] + 0.3)))
    for k in range(2, d):
        f = f + (0.3 + 0.2 * k) * X[:, k] ** 2
    Y = ((f - f.mean()) / f.std())[:, None]
    return X, Y


def build_model(n, d, seed, device):
    from dgp_amd import dgp, kernel, combine
    X, Y = synthetic(n, d)
    layers = combine([kernel(np.array([1.0]), name='matern2.5') for _ in range(d)],
                     [kernel(np.array([1.0]), name='matern2.5', scale_est=True, connect=np.arange(d))])
    np.random.seed(seed)
    return dgp(X, Y, layers, seed=seed, device=device), X, Y


def cpu_baseline(model, counts, ess_burn):
    """The oracle (numpy/scipy-LAPACK restatement of dgpsi's formulation, kind 'port') timed on this
    host on a bounded sample of the same workload: a few prior draws, ESS log-likelihoods and M-step
    objective evaluations at the bench sizes, scaled by the call counts one SI iteration makes
    (counted in the GPU run).  Reported beside the GPU number; never the thing measured."""
    from oracle import dgp_oracle as O
    import psutil
    rng = np.random.default_rng(1)
    l1, l2 = model.all_layer[0][0], model.all_layer[1][0]
    n = len(l1.output)
    reps_f, reps_l, reps_g = 3, 3, 3     # after one untimed call of each: a few seconds in all (the baseline must not outlast the GPU legs)
    # BLAS threads = physical cores.  The OpenBLAS inside numpy / scipy wheels is built with a fixed maximum (64 in
    # the wheels of this image): asking for more is clamped by the library, and the count it reports afterwards goes into the line.
    phys = psutil.cpu_count(logical=False) or 1
    limiter, blas_note = None, None
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=phys, user_api='blas')
    except Exception as exc:   # noqa: BLE001
        blas_note = 'threadpoolctl: %s' % exc

    def timed(f, reps):
        f()   # untimed: first-touch, BLAS thread pool start-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), ts

    # fmvn(scale*k_matrix()) of a first-layer node (imputation.py:63); log_likelihood_func of the second-layer node
    # (imputation.py:76,104); kernel.llik: objective + gradient, one cho_solve(n x n) per parameter
    t_fmvn, all_f = timed(lambda: O.fmvn(l1.scale[0] * O.k_matrix(l1._X(), l1.length, l1.nugget[0], l1.name), rng.standard_normal(n)), reps_f)
    t_ll, all_l = timed(lambda: O.log_likelihood(l2._X(), l2.output, l2.length, l2.scale, l2.nugget[0], l2.name), reps_l)
    t_llik, all_g = [], []
    for nd in (l1, l2):
        t, ts = timed(lambda nd=nd: O.nll_grad(nd.log_t(), nd._X(), nd.output, nd.name, nd.scale, nd.nugget[0], nd.nugget_est, nd.scale_est,
                                               nd.prior_name, nd.prior_coef), reps_g)
        t_llik.append(t)
        all_g += ts
    sweeps = ess_burn + 1
    n_l1 = len(model.all_layer[0])
    per_iter = (sweeps * n_l1 * t_fmvn                                       # reference refactors every sweep
                + (sweeps + counts['proposals_per_iter']) * t_ll              # threshold + proposals
                + counts['llik_l1_per_iter'] * t_llik[0] + counts['llik_l2_per_iter'] * t_llik[1])
    threads = None
    try:   # what the BLAS behind numpy / scipy actually runs with ("nothing verifies the 128 threads")
        from threadpoolctl import threadpool_info
        threads = [dict(api=i.get('user_api'), lib=i.get('internal_api'), threads=i.get('num_threads')) for i in threadpool_info()]
    except Exception:   # noqa: BLE001
        pass
    blas_threads = max([t['threads'] for t in (threads or []) if t.get('api') == 'blas'] or [phys])
    if blas_threads < phys and blas_note is None:
        blas_note = ('asked the BLAS for %d threads (= physical cores), it runs %d: the OpenBLAS built into the numpy / scipy wheels has a '
                     'compile-time maximum' % (phys, blas_threads))
    if limiter is not None:
        limiter.restore_original_limits()
    return dict(value=1.0 / per_iter, unit='SI it/s', cores=int(blas_threads), physical_cores=phys, kind='port',
                threadpools=threads, blas_threads_note=blas_note,
                sample=('one untimed call, then the median of %d fmvn, %d log_likelihood_func and 2x%d llik calls at n=%d (%.1f s of timed '
                        'CPU work), scaled by the per-iteration call counts of the GPU run: %d sweeps x %d fmvn, %.1f+%d log-liks, '
                        '%.1f/%.1f llik calls (layer 1/2)'
                        % (reps_f, reps_l, reps_g, n, sum(all_f) + sum(all_l) + sum(all_g), sweeps, n_l1,
                           counts['proposals_per_iter'], sweeps, counts['llik_l1_per_iter'], counts['llik_l2_per_iter'])),
                seconds_per_call=dict(fmvn=t_fmvn, log_likelihood_func=t_ll, llik_layer1=t_llik[0], llik_layer2=t_llik[1]))


def _cpu_predict_points(job):
    """One worker of cpu_baseline_predict (a spawned process, BLAS pinned to one thread: the reference's link_gp is a numba
    prange over the test points, functions.py:396-430 -- one core per point)."""
    import os
    os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
    import time as _t
    import numpy as _np
    from oracle import dgp_oracle as O
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:   # noqa: BLE001
        pass
    x, path = job
    import pickle as _pk
    with open(os.path.join(path, 'meta.pkl'), 'rb') as f:
        meta = _pk.load(f)
    arr = lambda name: _np.load(os.path.join(path, name + '.npy'), mmap_mode='r')   # noqa: E731  (memory-mapped: ONE copy of the model's arrays for all workers)
    l1 = [dict(nd, W=arr('l1_W_%d' % k), Rinv=arr('l1_Rinv_%d' % k), Rinv_y=arr('l1_ry_%d' % k)) for k, nd in enumerate(meta['l1'])]
    l2 = dict(meta['l2'], W=arr('l2_W'), Wg=(arr('l2_Wg') if meta['l2_has_Wg'] else None), Rinv=arr('l2_Rinv'), Rinv_y=arr('l2_ry'))
    t0 = _t.perf_counter()
    m = _np.empty((len(x), len(l1)))
    v = _np.empty((len(x), len(l1)))
    for k, nd in enumerate(l1):   # functions.gp (functions.py:379-394) of every first-layer node
        m[:, k], v[:, k] = O.gp_predict(x[:, nd['cols']], nd['W'], nd['Rinv'], nd['Rinv_y'], nd['scale'], nd['length'], nd['nugget'], nd['name'])
    z = None if l2['Wg'] is None else x[:, l2['gcols']]   # the output node's deterministic global inputs (connect)
    mu, var = O.link_gp_predict(m, v, z, l2['W'], l2['Wg'], l2['Rinv'], l2['Rinv_y'], l2['scale'], l2['length'], l2['nugget'], l2['name'])
    return _t.perf_counter() - t0, float(mu.sum()), float(var.sum())


def cpu_baseline_predict(model, d, imputations, points_per_worker=1):
    """The predict half of the metric on the host: the oracle's gp_predict (first layer) + link_gp_predict (Matern-2.5 IJ with the
    reference's closed forms, functions.py:379-430,453-494) at n = 2000 for ONE imputation (the model's current latent state;
    compute_stats = emulator.__init__ is not timed, as SURVEY 8(d) defines the metric) on a few test points per worker process,
    one process per core up to 32 -- the reference parallelises link_gp over the test points -- scaled to `imputations`
    imputations per point.  Reported beside the GPU number; never the thing measured."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    import psutil
    from oracle import dgp_oracle as O
    phys = psutil.cpu_count(logical=False) or 1
    # One process per core up to 32.  One per PHYSICAL core was tried (profiles/r06_bench_default_cpu128.json): 128 processes of this numpy port take 177 s
    # per point where 32 take 12 s -- 0.072 against 0.27 pts/s in aggregate (every process streams n x n temporaries: the host's memory system, not its cores, is what they
    # share) -- and the leg alone then lasts three minutes.  So: the faster configuration, within the contract's 10-30 s of CPU work; `physical_cores` is reported beside it.
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = phys
    workers = max(1, min(32, phys, avail))
    try:   # (a worker holds ~0.4 GB of n x n temporaries of the oracle's IJ at n = 2000: never more than a quarter of the free memory in all)
        workers = max(1, min(workers, int(0.25 * psutil.virtual_memory().available / 0.4e9)))
    except Exception:   # noqa: BLE001
        pass
    l1 = []
    for nd in model.all_layer[0]:
        Xn = np.ascontiguousarray(nd._X())
        st = O.compute_stats(Xn, nd.output[:, 0], nd.length, nd.nugget[0], nd.name, Xn.shape[1])
        l1.append(dict(cols=np.asarray(nd.input_dim) if nd.input_dim is not None else np.arange(Xn.shape[1]),
                       W=Xn, Rinv=st['Rinv'], Rinv_y=st['Rinv_y'], scale=nd.scale, length=nd.length, nugget=nd.nugget, name=nd.name))
    nd = model.all_layer[1][0]
    W2 = np.ascontiguousarray(nd._X())
    st = O.compute_stats(W2, nd.output[:, 0], nd.length, nd.nugget[0], nd.name, nd._input.shape[1])
    nloc = nd._input.shape[1]
    l2 = dict(W=np.ascontiguousarray(W2[:, :nloc]), Wg=None if nd._global_input is None else np.ascontiguousarray(W2[:, nloc:]),
              gcols=None if nd.connect is None else np.asarray(nd.connect), Rinv=st['Rinv'], Rinv_y=st['Rinv_y'], scale=nd.scale, length=nd.length,
              nugget=nd.nugget, name=nd.name)
    xt = np.random.default_rng(11).uniform(size=(workers * points_per_worker, d))
    import pickle
    import shutil
    import tempfile
    shm = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else None
    path = tempfile.mkdtemp(prefix='dgp_amd_cpu_baseline_', dir=shm)   # (~0.2 GB: the six nodes' R^-1; removed below)
    try:
        with open(os.path.join(path, 'meta.pkl'), 'wb') as f:
            pickle.dump(dict(l1=[{k: v for k, v in nd.items() if k not in ('W', 'Rinv', 'Rinv_y')} for nd in l1],
                             l2={k: v for k, v in l2.items() if k not in ('W', 'Wg', 'Rinv', 'Rinv_y')}, l2_has_Wg=l2['Wg'] is not None), f)
        save = lambda name, a: np.save(os.path.join(path, name + '.npy'), np.ascontiguousarray(a))   # noqa: E731
        save('l2_W', l2['W']); save('l2_Rinv', l2['Rinv']); save('l2_ry', l2['Rinv_y'])
        if l2['Wg'] is not None:
            save('l2_Wg', l2['Wg'])
        for k, nd in enumerate(l1):
            save('l1_W_%d' % k, nd['W']); save('l1_Rinv_%d' % k, nd['Rinv']); save('l1_ry_%d' % k, nd['Rinv_y'])
        jobs = [(xt[w * points_per_worker:(w + 1) * points_per_worker], path) for w in range(workers)]
        t0 = time.perf_counter()
        with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context('spawn')) as ex:   # (spawn: this process holds a HIP context)
            res = list(ex.map(_cpu_predict_points, jobs))
        wall = time.perf_counter() - t0
    finally:
        shutil.rmtree(path, ignore_errors=True)
    busy = max(r[0] for r in res)            # the slowest worker's compute time (process start-up and pickling excluded)
    pts = len(xt)
    finite = bool(np.all(np.isfinite([r[1] for r in res])) and np.all(np.isfinite([r[2] for r in res])))
    return dict(value=pts / busy / imputations, unit='pts/s', cores=workers, physical_cores=phys, kind='port', imputations=imputations,
                seconds_per_point_imputation_one_core=float(np.mean([r[0] for r in res])) / points_per_worker, finite=finite,
                note=('the reference evaluates link_gp in numba (functions.py:396-430, a prange over the test points); numba is absent from this image and cannot travel, so this is '
                      'the oracle\'s numpy restatement of the same closed forms, one process per core up to 32 (128 processes were slower in aggregate: profiles/'
                      'r06_bench_default_cpu128.json) -- one to two orders of magnitude slower per core than compiled code: a reported baseline, not a target'),
                sample=('%d worker processes x %d test points, one imputation each (gp_predict of the %d first-layer nodes + link_gp_predict of the '
                        'Matern-2.5 output node at n=%d, one BLAS thread per process as the reference\'s prange over test points); slowest worker %.1f s, '
                        '%.1f s wall with process start-up; scaled to %d imputations per point'
                        % (workers, points_per_worker, len(l1), len(W2), busy, wall, imputations)))


def potrf_table(eng, torch, n=2000):
    """The factorisation alone at the bench size, by batch (HIP events on the engine's stream, minimum of 5): ms and algorithmic
    TFLOP/s (n^3/3 per matrix, n^3 with the fused inverse) -- the headline roofline fraction averages over whatever batch sizes
    the sampled iterations held (half of all M-step rounds carry one matrix), this table does not."""
    Np = eng.padded_dim(n)
    rng = np.random.default_rng(1)
    out = {}
    ev0, ev1 = eng.event(), eng.event()
    for B in (1, 2, 3, 4, 6, 10, 12):
        X = eng.tensor(rng.uniform(size=(B, n, 5)))
        y = eng.tensor(rng.normal(size=n))
        A, T, S = eng.empty(B, Np, Np), eng.empty(B, Np, Np), eng.empty(B, Np, Np)
        work = eng.potrf_workspace(n, B)
        tf, tv = [], []
        for rep in range(5):
            eng.kmatrix('matern2.5', X, None, None, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
            eng.record(ev0); eng.potrf(n, A, batch=B, work=work); eng.record(ev1)
            torch.cuda.synchronize()
            tf.append(eng.elapsed_ms(ev0, ev1))
            eng.kmatrix('matern2.5', X, None, None, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
            eng.record(ev0); eng.potrf_inv(n, A, T, S, batch=B, work=work); eng.record(ev1)
            torch.cuda.synchronize()
            tv.append(eng.elapsed_ms(ev0, ev1))
        f, v = min(tf[1:]), min(tv[1:])
        out['B=%d' % B] = dict(potrf_ms=f, potrf_tflops=B * n ** 3 / 3 / f / 1e9, potrf_frac=B * n ** 3 / 3 / f / 1e9 / F64_PEAK_TFLOPS,
                               potrf_inv_ms=v, potrf_inv_tflops=B * n ** 3 / v / 1e9, potrf_inv_frac=B * n ** 3 / v / 1e9 / F64_PEAK_TFLOPS)
        del A, T, S
    out['note'] = ('n=%d; one call = one potrf_mega_kernel launch (+ its memset / copy-out), minimum of 4; one matrix is bound by the 31-step '
                   'pivot chain (12.6 us per step alone, 13.9 with the tasks of the inverse around it), an M-step round holds 1-5 matrices with the inverse, a speculative ESS batch 10 or 6 without' % n)
    return out


def vecchia_leg(eng, torch, n=50000, d=8, m=25, B=6):
    """SURVEY 8(d), Vecchia rows at BASELINE configs[3]'s shape (vecchia.py:20-109,164-242): ordered m-NN search, log-likelihood
    rows (alone and as a speculative batch), objective + gradient rows, sparse-factor rows; HIP-event timed, ~2 s in all."""
    rng = np.random.default_rng(7)
    X = rng.uniform(size=(n, d))
    y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 + 0.1 * rng.normal(size=n)
    length = np.array([0.8])
    dX, dy, ones = eng.tensor(X), eng.tensor(y), eng.tensor(np.ones(n))
    XB = dX.unsqueeze(0).repeat(B, 1, 1).contiguous()
    xs = eng.tensor(X / length)

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        with eng.stream():
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(reps):
                fn()
            e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps   # ms
    t_nn = timed(lambda: eng.nn_ordered(xs, m), 3)
    NN = eng.nn_ordered(xs, m)
    t1 = timed(lambda: eng.vecchia_llik('matern2.5', dX, dy, NN, length, 1e-4, ones), 20)
    tb = timed(lambda: eng.vecchia_llik_batch('matern2.5', XB, dy, NN, length, 1e-4, ones), 20)
    t2 = timed(lambda: eng.vecchia_nllik('matern2.5', dX, dy, NN, length, 1e-4, ones, True), 20)
    t3 = timed(lambda: eng.vecchia_lmatrix('matern2.5', dX, NN, length, 1e-4), 20)
    gather = n * (m + 1) * (d + 1) * 8.0   # bytes a likelihood evaluation gathers through the neighbour array (coordinates + output per neighbour)
    # the query form at cfg4's prediction shape (vecchia.py:20-59: pred_m = 50 nearest training points of every test point)
    Mq, pm = 100000, 50
    xq = eng.tensor(rng.uniform(size=(Mq, d)) / length)
    t_q = timed(lambda: eng.nn_query(xq, xs, pm), 3)
    # brute force = 2 n^2 D flops (ordered: every point against all others; the kernels evaluate the n^2/2 earlier ones), queries 2 M n D;
    # priced against the f64 vector peak (the distance chains are f64 fma)
    nn_roof = dict(bound='f64 VALU', unit='TFLOP/s', peak=F64_PEAK_TFLOPS,
                   ordered=dict(ms=t_nn, flops=2.0 * n * n * d, achieved=2.0 * n * n * d / t_nn / 1e9, frac=2.0 * n * n * d / t_nn / 1e9 / F64_PEAK_TFLOPS,
                                executed_flops=1.0 * n * n * d, what='%d-NN among the earlier points of %d, D=%d (nn_scan_kernel + nn_merge_kernel)' % (m, n, d)),
                   query=dict(ms=t_q, flops=2.0 * Mq * n * d, achieved=2.0 * Mq * n * d / t_q / 1e9, frac=2.0 * Mq * n * d / t_q / 1e9 / F64_PEAK_TFLOPS,
                              what='%d nearest of %d points for %d queries, D=%d (nn_tau / nn_collect / nn_pick: filter, then select)' % (pm, n, Mq, d)),
                   note='algorithmic flops as SURVEY 8(d) counts them (2 per dimension and pair: subtract, multiply-add); the ordered search needs only the pairs (i, j < i): '
                        'executed = half of that.  Beside the distances the kernels keep K sorted (distance, index) pairs per query -- insertion work that the flop count does not see')
    return dict(shape='n=%d, d=%d, m=%d, Matern-2.5' % (n, d, m), ordered_nn_ms=t_nn, nn_query_ms=t_q, roofline_nn=nn_roof,
                llik_ms=t1, llik_rows_per_s=n / t1 * 1e3, llik_gather_GBs=gather / t1 / 1e6,
                llik_batch=dict(candidates=B, ms=tb, rows_per_s=B * n / tb * 1e3, gather_GBs=B * gather / tb / 1e6),
                nllik_ms=t2, nllik_rows_per_s=n / t2 * 1e3, lmatrix_ms=t3, lmatrix_rows_per_s=n / t3 * 1e3,
                bound=pmc_note('vecchia_row4_kernel', 'f64 VALU issue (~0.79 of peak by in-kernel stamps, profiles/r04_vecchia_row_kernel_phases.txt)')
                + ': the gather is %.1f MB per evaluation, a few per cent of the HBM roofline' % (gather / 1e6))


def sexp_pair_leg(eng, torch, n=5000, Dw=10, Dz=10, M=2048):
    """SURVEY 8(d), link_gp with the squared-exponential kernel at BASELINE configs[2]'s second-layer shape (functions.py:396-451: n = 5000, ten
    uncertain local inputs + ten deterministic global ones, 2048 test points = one launch of the pair kernel): R^-1 from the device's own
    factorisation of a synthetic node, then the predictor call with HIP events around the pair kernel (linkgp_Jsexp2_kernel)."""
    rng = np.random.default_rng(5)
    W, Wg = rng.normal(size=(n, Dw)), rng.uniform(size=(n, Dz))
    y = np.sin(W[:, 0]) + Wg[:, 1] ** 2 + 0.1 * rng.normal(size=n)
    length, scale, nugget = np.array([2.5]), 1.3, 1e-4
    Np = eng.padded_dim(n)
    A = eng.kmatrix('sexp', eng.tensor(np.concatenate((W, Wg), 1)), None, None, length, nugget, full=False, Y=eng.tensor(y))
    _, info = eng.potrf(n, A)
    Ainv = eng.empty(Np, Np)
    eng.potri(n, A, Ainv, 1, eng.potrf_workspace(n, 1))
    ry = (-Ainv[n, :n]).contiguous()
    m, v, z = eng.tensor(rng.normal(size=(M, Dw))), eng.tensor(rng.uniform(0.01, 0.4, size=(M, Dw))), eng.tensor(rng.uniform(size=(M, Dz)))
    Wd, Wgd = eng.tensor(W), eng.tensor(Wg)
    call = lambda: eng.linkgp_predict('sexp', m, v, z, Wd, Wgd, length, Ainv, Np, ry, scale, nugget)   # noqa: E731
    mu, var = call()
    torch.cuda.synchronize()
    ok = bool(int(info.cpu().numpy()[0]) == 0 and torch.isfinite(mu).all() and torch.isfinite(var).all())
    eng.prof_enable('linkgp_j')
    for _ in range(3):
        call()
    k, ms, pairs = eng.prof_collect()
    if not k:
        return dict(error='the pair kernel was not launched')
    calls = 3
    pair_s = pairs / (ms * 1e-3)                    # pair evaluations per second (one exponential each): what the kernel executes, the n (n + 1) / 2 lower pairs
    el_s = 2.0 * pair_s                             # J elements per second as the reference evaluates them (all n^2: SURVEY 8(d)'s unit)
    # Executed double-precision work per pair evaluation (ISA of the loop, profiles/r05_sexp_pair_kernel.txt: per test point and wave -- 16 x 64 pairs -- 12 MFMAs
    # 16x16x4 for the (Dw + 2)-wide exponent and 166 f64 VALU instructions, the table exponential among them): 24 MFMA flops + 10.4 f64 VALU lane-instructions, an fma
    # counted as 2 flops.  SURVEY 8(d)'s own model (M n^2 (8 D flops + 1 exp) for the reference's form) is reported beside it: the kernel evaluates half the
    # elements (symmetry) and forms the exponent in 2 (Dw + 2) flops instead of 8 Dw, so that figure exceeds the peak and is no utilisation.
    flop_pair = 24.0 + 2.0 * 166.0 * 64.0 / 1024.0
    ach = pair_s * flop_pair / 1e12
    return dict(bound='f64 VALU + MFMA (shared double-precision units)', kernel='linkgp_Jsexp2_kernel (SExp pair phase: exponent on f64 MFMA, one table exponential per pair)',
                shape='n=%d, %d uncertain + %d deterministic inputs, %d test points' % (n, Dw, Dz, M), launches=k, launches_per_call=k / calls, avg_launch_us=1e3 * ms / k,
                ms_per_2048_points=(ms / calls) * 2048.0 / M, pair_evaluations_per_s=pair_s, J_elements_per_s=el_s,
                achieved=ach, peak=F64_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / F64_PEAK_TFLOPS,
                flop_convention='executed: %.1f double-precision flops per pair evaluation (24 on MFMA + 10.4 f64 VALU instructions x 2)' % flop_pair,
                survey_model_tflops=el_s * (8.0 * Dw + 20.0) / 1e12, traffic=None, finite=ok,
                note=pmc_note('linkgp_Jsexp2_kernel', 'counters of this kernel: profiles/r05_pmc_kernels.txt'))


def strong_leg_cfg3(dd, torch, local, dev, world, n=5000, d=10, q=3, S=16, M=2048):
    """BASELINE configs[2] shape (2-layer DGP, d=10 in / 3 out, n=5000, default structure: SExp): S imputations IN ALL,
    sharded over the ranks (emulation.py:701-779 per imputation; one all-reduce of the two moment arrays), M test points."""
    from dgp_amd import dgp, emulator
    rng = np.random.default_rng(2026)
    X = rng.uniform(size=(n, d))
    Y = np.stack([np.sin(1.0 / ((0.7 * X[:, 0] + 0.3) * (0.7 * X[:, 1] + 0.3))) + (0.2 + 0.1 * j) * (X[:, 2 + j:] ** 2).sum(1) for j in range(q)], 1)
    Y = (Y - Y.mean(0)) / Y.std(0)
    np.random.seed(1)
    model = dgp(X, Y, seed=1, device=local)          # the same model on every rank (same seeds)
    model.train(N=1, ess_burn=10, disable=True)
    emu = emulator(model.estimate(burnin=0), N=S, seed=3, device=local)   # N/world imputations on this rank
    xt = rng.uniform(size=(M, d))
    emu.predict(xt[:64])
    dd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mu, var = emu.predict(xt)
    torch.cuda.synchronize()
    dd.barrier()
    t = dd.allreduce_max_scalar(time.perf_counter() - t0, dev)
    return dict(what='cfg3 shape: n=%d, d=%d in / %d out, %d imputations in all over %d ranks, %d test points' % (n, d, q, S, world, M),
                seconds=t, point_imputations_per_s=M * S / t, finite=bool(np.all(np.isfinite(mu)) and np.all(np.isfinite(var))))


def strong_leg_cfg4(dd, torch, local, dev, world, n=50000, d=8, m=25, its=2):
    """BASELINE configs[3] shape (Vecchia DGP, n=50000, d=8, m=25): one model trained by all ranks with the rows of
    vecchia_llik / vecchia_nllik split over them (vecchia.py:165-242's prange; dist.split_training(rows=True))."""
    from dgp_amd import dgp
    rng = np.random.default_rng(7)
    X = rng.uniform(size=(n, d))
    f = np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) + X[:, 2] ** 2 + 0.3 * X[:, 3:].sum(1)
    Y = ((f - f.mean()) / f.std())[:, None]
    dd.split_training(nodes=False, rows=True)
    np.random.seed(1)                                   # (the Vecchia ordering is a numpy.random.permutation: the same on every rank)
    model = dgp(X, Y, vecchia=True, m=m, seed=1, device=local)
    model.train(N=1, ess_burn=10, disable=True)
    dd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.train(N=its, ess_burn=10, disable=True)
    torch.cuda.synchronize()
    dd.barrier()
    t = dd.allreduce_max_scalar(time.perf_counter() - t0, dev)
    dd.split_training(rows=False)
    return dict(what='cfg4 shape: Vecchia DGP n=%d, d=%d, m=%d, rows split over %d ranks' % (n, d, m, world), steps=its,
                ms_per_step=1e3 * t / its, si_it_per_s=its / t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=150)   # ~5.5 s of timed region: long enough for the driver's SMI sampler
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--n', type=int, default=2000)
    ap.add_argument('--d', type=int, default=5)
    ap.add_argument('--ess-burn', type=int, default=10)
    ap.add_argument('--predict-points', type=int, default=16384)
    ap.add_argument('--predict-seconds', type=float, default=3.0, help='the prediction leg repeats its predict() call until it has run this long')
    ap.add_argument('--min-gpu-seconds', type=float, default=0.0,
                    help='telemetry padding, off by default: after everything that is measured, keep stepping (untimed, unreported) until the '
                         'GPU legs have lasted this long in all, for an external sampler of device activity that needs a longer run')
    ap.add_argument('--no-strong-legs', action='store_true', help='N > 1: skip the strong-scaling legs (cfg3 prediction with the imputations sharded, cfg4 training with the Vecchia rows split)')
    ap.add_argument('--imputations', type=int, default=10)
    ap.add_argument('--sustained-steps', type=int, default=100, help='iterations of the train(N) call behind the timed region (0: skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-predict', action='store_true')
    ap.add_argument('--no-split-leg', action='store_true', help='N > 1: skip the informational leg that trains one model with the M-step nodes split over the ranks')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="process-group backend for N > 1; 'gloo' lets several ranks share one GPU (functional check of the "
                         "multi-rank path on a one-GPU box: LOCAL_RANK modulo the device count)")
    ap.add_argument('--prof-kernel', default='syrk',
                    help="kernel class timed with HIP events for the roofline ('syrk' = the fused block-step kernel "
                         "of the factorisation, the dominant kernel; 'lauum', 'trtri', 'kmatrix', ...)")
    args = ap.parse_args()

    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (before anything touches the GPU: a
        # child process, never an exec) and pass their output / exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr',
               '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world_env != max(1, args.gpus):
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run --nproc-per-node %d' % (args.gpus, world_env, args.gpus))

    # (the pool's host driver supports dmabuf IPC only: RCCL between processes fails with `hipIpcGetMemHandle: invalid argument` otherwise; the
    #  launcher's environment carries it already -- set before the HIP runtime starts in case it does not)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    from dgp_amd import dist as dd
    from dgp_amd import kernel_class
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    forced = world == 1 and os.environ.get('DGPAMD_DIST_FORCE') == '1'   # a process group of ONE rank: RCCL first contact on a one-GPU box
    if world > 1 or forced:
        dd.init_from_env(args.backend)
    shared_device = False
    c_stdio = None
    try:   # RCCL prints its version banner through C stdio, which a pipe buffers until exit: flushed before the result line is written,
        import ctypes   # so that the JSON line stays the LAST line of stdout
        c_stdio = ctypes.CDLL(None)
    except OSError:
        pass
    if args.backend == 'gloo':
        shared_device = world > torch.cuda.device_count()
        local = local % max(1, torch.cuda.device_count())
        if shared_device:
            # several ranks on ONE GPU (the functional check of the N > 1 flow on a one-GPU box, never a deployment): the
            # one-launch factorisation expects its workgroups co-resident and, time-sliced against another process, runs
            # into its hand-off bound (DgpAmdError; INTEGRATION.md, operational notes) -- the per-block-step kernel does not spin
            os.environ['DGPAMD_POTRF_MODE'] = '0'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local) if args.backend == 'nccl' else None   # (gloo reduces host tensors)

    model, X, Y = build_model(args.n, args.d, 100 + rank, local)
    eng = model.engine

    # count M-step objective evaluations per layer (for the CPU baseline's scaling)
    calls = {'l1': 0, 'l2': 0}
    layer_of = {id(nd): ('l1' if l == 0 else 'l2') for l, layer in enumerate(model.all_layer) for nd in layer}
    orig_finish = kernel_class.kernel._llik_finish

    def counted(self, host):
        calls[layer_of[id(self)]] += 1
        return orig_finish(self, host)
    kernel_class.kernel._llik_finish = counted

    # ... and the lock-step optimiser rounds of every M-step: the reference's L-BFGS-B runs take between 6 and 45 rounds per
    # iteration along a training path (the slowest node's evaluations; maxfun = 45), so the work inside K timed steps varies
    # by +-10 % at K = 20 between builds whose objectives differ in the last bit -- the line reports it beside the rate
    from dgp_amd import mstep as mstep_mod
    rounds = {'n': 0}
    orig_lock = mstep_mod.minimize_lockstep

    # ---- the step accounts for itself (host clock stamps at the phase boundaries and around every objective round of the lock-step M-step,
    #      three pre-created HIP events per step on the engine's stream; ~10 us of stamps per 30-ms step, no profiler) ----
    acct = dict(on=False, i_host=0.0, m_host=0.0, evaluate=0.0, turn=0.0, head=0.0, tail=0.0, rounds_by_batch={}, eval_by_batch={}, events=[], k=0)

    def counted_lock(problems, evaluate, *a, **k):
        if not acct['on'] or k.get('groups', 1) != 1:
            r = orig_lock(problems, evaluate, *a, **k)
            rounds['n'] += r
            return r
        st = dict(last=None, t0=time.perf_counter(), t=None)

        def begin():
            t = time.perf_counter()
            if st['last'] is None:
                acct['head'] += t - st['t0']          # the optimisers' first advance
            else:
                acct['turn'] += t - st['last']        # host turn-around between two rounds: results absorbed, optimisers advanced
            st['t'] = t

        def end(req):
            t2 = time.perf_counter()
            acct['evaluate'] += t2 - st['t']          # one round: hyper-parameters in, K assembly + factorisation + inverse + reductions, results on the host
            B = len(req)
            acct['rounds_by_batch'][B] = acct['rounds_by_batch'].get(B, 0) + 1
            acct['eval_by_batch'][B] = acct['eval_by_batch'].get(B, 0.0) + (t2 - st['t'])
            st['last'] = t2

        def timed_evaluate(req):
            begin()
            out = evaluate(req)
            end(req)
            return out
        if hasattr(evaluate, 'launch'):   # (the first round in two halves: the caller's deferred host work runs between them, inside this round's time)
            def t_launch(req, slot=0):
                begin()
                return evaluate.launch(req, slot)

            def t_collect(req, token):
                out = evaluate.collect(req, token)
                end(req)
                return out
            timed_evaluate.launch, timed_evaluate.collect = t_launch, t_collect
            if hasattr(evaluate, 'abandon'):
                timed_evaluate.abandon = evaluate.abandon
        r = orig_lock(problems, timed_evaluate, *a, **k)
        if st['last'] is not None:
            acct['tail'] += time.perf_counter() - st['last']
        rounds['n'] += r
        return r
    mstep_mod.minimize_lockstep = counted_lock

    # one step = one iteration of dgp.train's own loop (dgp_amd.dgp._si_iteration <- dgp.py:1377-1398: imputer.sample, then the M-step; iteration numbers >= 2:
    # the first iteration of a train() call initialises the scales).  The stamps at the I / M boundary ride on the two calls it makes.
    it_no = {'i': 1}
    imp_sample = model.imp.sample

    def stamped_sample(*a, **k):
        r = imp_sample(*a, **k)
        if acct['on']:
            acct['t1'] = time.perf_counter()
            eng.record(acct['events'][acct['k']][1])
        return r
    model.imp.sample = stamped_sample

    def step():
        it_no['i'] += 1
        if not acct['on']:
            model._si_iteration(it_no['i'], args.ess_burn)
            return
        e0, e1, e2 = acct['events'][acct['k']]
        t0 = time.perf_counter()
        eng.record(e0)
        model._si_iteration(it_no['i'], args.ess_burn)
        eng.record(e2)
        t2 = time.perf_counter()
        acct['k'] += 1
        acct['i_host'] += acct['t1'] - t0
        acct['m_host'] += t2 - acct['t1']

    t_gpu0 = time.perf_counter()
    for _ in range(args.warmup):
        step()
    calls['l1'] = calls['l2'] = 0
    rounds['n'] = 0
    st0 = dict(model.imp.stats)
    acct['events'] = [(eng.event(), eng.event(), eng.event()) for _ in range(args.steps)]
    dd.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = eng.event(), eng.event()
    eng.record(ev0)
    acct['on'] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.record(ev1)
    torch.cuda.synchronize()
    dd.barrier()
    wall = time.perf_counter() - t0
    acct['on'] = False
    wall = dd.allreduce_max_scalar(wall, dev if world > 1 else None)
    steps_total = args.steps * world
    value = steps_total / wall
    st1 = model.imp.stats
    upd = max(1, st1['updates'] - st0['updates'])
    counts = dict(proposals_per_iter=(st1['proposals'] - st0['proposals']) / args.steps,
                  batches_per_update=(st1['batches'] - st0['batches']) / upd,
                  llik_l1_per_iter=calls['l1'] / args.steps, llik_l2_per_iter=calls['l2'] / args.steps,
                  mstep_rounds_per_iter=rounds['n'] / args.steps)

    # ---- step_split: where the timed region's wall time went (per step, ms) ----
    step_split = None
    if rank == 0:
        K = float(args.steps)
        gi = sum(eng.elapsed_ms(a, b) for a, b, _ in acct['events']) / K
        gm = sum(eng.elapsed_ms(b, c) for _, b, c in acct['events']) / K
        gap = sum(eng.elapsed_ms(acct['events'][i][2], acct['events'][i + 1][0]) for i in range(args.steps - 1)) / K
        ms = lambda x: 1e3 * x / K   # noqa: E731
        other_m = acct['m_host'] - acct['evaluate'] - acct['turn'] - acct['head'] - acct['tail']
        parts = dict(istep_ms=ms(acct['i_host']), mstep_rounds_device_calls_ms=ms(acct['evaluate']), mstep_host_turnaround_between_rounds_ms=ms(acct['turn']),
                     mstep_first_advance_ms=ms(acct['head']), mstep_after_last_round_ms=ms(acct['tail']), mstep_setup_and_diagnostics_ms=ms(other_m),
                     outside_ms=1e3 * wall / K - ms(acct['i_host'] + acct['m_host']))
        rb = {int(b): c / K for b, c in sorted(acct['rounds_by_batch'].items())}
        eb = {int(b): 1e3 * acct['eval_by_batch'][b] / acct['rounds_by_batch'][b] for b in sorted(acct['rounds_by_batch'])}
        step_split = dict(parts_ms_per_step=parts, sum_ms=sum(parts.values()), ms_per_step=1e3 * wall / K,
                          mstep_rounds_per_step_by_matrices=rb, mstep_round_ms_by_matrices=eb,
                          mstep_objective_evaluations_per_step=sum(b * c for b, c in rb.items()),
                          gpu_timeline_ms_per_step=dict(istep_span=gi, mstep_span=gm, between_steps=gap, total=gi + gm + gap),
                          istep=dict(sweeps=args.ess_burn + 1, updates_per_step=upd / K, proposals_per_step=counts['proposals_per_iter'],
                                     speculative_batches_per_step=(st1['batches'] - st0['batches']) / K),
                          how='host clock at the phase boundaries and around every objective round of the lock-step driver (one round = one dgpamd_llik_batch call: '
                              'hyper-parameters in, K assembly + potrf_inv + gradient reductions of the round\'s matrices, results on the host), summed over the '
                              'timed region; the parts are disjoint and cover the region, so sum_ms = ms_per_step up to clock reads.  gpu_timeline: HIP events recorded on the '
                              'engine\'s stream at the same boundaries (when the DEVICE passed them).  Kernel-level times of the same build: roofline (potrf launches), '
                              'potrf_table (by batch), profiles/r06_idle_gaps.txt (rocprofv3 trace of this command)')

    # ---- roofline of the dominant kernel: HIP events around each of its launches, same steps ----
    roof = None
    if rank == 0 and args.prof_kernel != 'none':
        engines = [eng]
        for e in engines:
            e.prof_enable(args.prof_kernel)
        for _ in range(max(1, min(12, args.steps))):   # (twelve extra steps: an iteration holds 6 to 45 optimiser rounds, four were a noisy sample)
            step()
        tot_n, tot_ms, tot_w = 0, 0.0, 0.0
        for e in engines:
            k, ms, w = e.prof_collect()
            tot_n, tot_ms, tot_w = tot_n + k, tot_ms + ms, tot_w + w
        # An EMPTY event pair reports a few microseconds by itself, but that cost does not add to a bracketed kernel: the
        # raw event average agrees with rocprofv3's per-kernel average of the same command (32.5 vs 33.7 us in
        # profiles/r01_bench_train_predict_kernel_stats.txt), the "overhead removed" figure does not -- the raw one is reported.
        ev_us = float(np.median([eng.prof_event_overhead_us() for _ in range(5)]))
        corrected_ms = max(tot_ms - tot_n * ev_us * 1e-3, 0.25 * tot_ms)
        if tot_n:
            if args.prof_kernel == 'kmatrix':
                ach = tot_w / (tot_ms * 1e-3) / 1e9
                roof = dict(bound='hbm', kernel='kmatrix_kernel', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s',
                            frac=ach / HBM_PEAK_GBS, traffic=None)
            else:
                ach = tot_w / (tot_ms * 1e-3) / 1e12
                roof = dict(bound='mfma', kernel={'syrk': 'potrf_mega_kernel (one launch = one batched factorisation, with or without the fused inverse)'}.get(args.prof_kernel, 'tile_gemm_kernel<%s>' % args.prof_kernel), achieved=ach,
                            peak=F64_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / F64_PEAK_TFLOPS, traffic=None)
            pmc = next((q for q in (os.path.join(ROOT, 'profiles', 'r%02d_pmc_bench_potrf_kernel.json' % r) for r in (6, 5, 4, 2)) if os.path.exists(q)), '')
            if args.prof_kernel == 'syrk' and os.path.exists(pmc):   # HBM bytes per launch, measured offline with rocprofv3 --pmc
                with open(pmc) as f:
                    pj = json.load(f)
                roof['traffic'] = pj['hbm_bytes_per_launch']
                roof['traffic_source'] = ('profiles/' + os.path.basename(pmc) + ': ' + pj.get('how', 'FETCH_SIZE x2 + WRITE_SIZE per launch of the same kernel')
                                          + ' in separate rocprofv3 --pmc passes of `%s` (an earlier run of the same build, not this one)' % pj.get('command', '?'))
            roof.update(launches=tot_n, avg_launch_us=1e3 * tot_ms / tot_n, work_per_launch=tot_w / tot_n,
                        event_pair_overhead_us=ev_us, avg_launch_us_if_event_overhead_removed=1e3 * corrected_ms / tot_n,
                        note='algorithmic flops = n^3/3 per matrix (n^3 with the fused inverse), SURVEY 8(d); speculative batches '
                             'whose predicate turned them into no-ops are not counted (neither work nor time); under full f64 MFMA '
                             'load the chip holds ~1.85 of its 2.4 GHz, i.e. ~60 of the 78.6 TFLOP/s datasheet peak are attainable')
    # K assembly against the HBM roofline on the same steps (its own pass: one kernel class is timed at a time)
    roof_k = None
    if rank == 0 and args.prof_kernel != 'none':
        eng.prof_enable('kmatrix')
        step()
        k_n, k_ms, k_w = eng.prof_collect()
        if k_n:
            ach = k_w / (k_ms * 1e-3) / 1e9
            roof_k = dict(bound='hbm', kernel='kmatrix_kernel (lower tiles of the batched augmented buffers, as the training path assembles them)',
                          achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s', frac=ach / HBM_PEAK_GBS, traffic=None, launches=k_n,
                          avg_launch_us=1e3 * k_ms / k_n, bytes_per_launch=k_w / k_n,
                          note=pmc_note('kmatrix_kernel', 'this kernel is f64-VALU-bound, not HBM-bound (one entry per lane and ~70 VALU instructions '
                                        'per Matern entry, 22 of them the double-precision exp); the HBM fraction is reported because the contract asks for it'))
    # ---- the sustained figure: the reference's own entry point, train(N), over 100 iterations (the optimiser rounds per iteration range
    #      6-45 along a training path, so a 20-step window reads +-10 %; SURVEY 8(d) defines the metric on train(N)) ----
    sustained = None
    if args.sustained_steps > 0:
        r0 = rounds['n']
        dd.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train(N=args.sustained_steps, ess_burn=args.ess_burn, disable=True)
        torch.cuda.synchronize()
        dd.barrier()
        ts = dd.allreduce_max_scalar(time.perf_counter() - t0, dev if world > 1 else None)
        sustained = dict(si_it_per_s=args.sustained_steps * world / ts, steps=args.sustained_steps, seconds=ts,
                         mstep_rounds_per_iter=(rounds['n'] - r0) / args.sustained_steps,
                         what='dgp.train(N=%d, ess_burn=%d) behind the timed region and the roofline passes (same model, same chain)' % (args.sustained_steps, args.ess_burn))

    kernel_class.kernel._llik_finish = orig_finish

    # ---- K assembly alone against the HBM roofline, cache-defeating: full symmetric n = 5000, D = 10 (cfg3's node shape),
    #      200 MB per matrix, three output buffers in turn (600 MB > the 256-MB Infinity Cache), HIP events on the engine's stream
    roof_ks = None
    if rank == 0 and args.prof_kernel != 'none':
        roof_ks = {}
        nk, Dk = 5000, 10
        Xk = eng.tensor(np.random.default_rng(3).uniform(size=(nk, Dk)))
        outs = [eng.empty(nk, nk) for _ in range(3)]
        # ... and the same matrices as views of rows that start on 128-byte lines (5120 doubles per row): with n = ld = 5000 every second row
        # starts in the middle of a line and every tile edge splits lines between two workgroups (profiles/r05_kmatrix_hbm_roofline.txt)
        outs_al = [eng.empty(nk, 5120)[:, :nk] for _ in range(3)]
        lk = np.full(Dk, 0.9)
        for name, targets in (('sexp', outs), ('matern2.5', outs), ('sexp_rows_on_128B_lines', outs_al), ('matern2.5_rows_on_128B_lines', outs_al)):
            kind = name.split('_')[0]
            for o in targets:
                eng.kmatrix(kind, Xk, None, None, lk, 1e-6, out=o, full=True)
            torch.cuda.synchronize()
            eng.prof_enable('kmatrix')       # HIP events around every launch (the host cannot issue 45-us kernels back to back)
            reps = 30
            for r in range(reps):
                eng.kmatrix(kind, Xk, None, None, lk, 1e-6, out=targets[r % 3], full=True)
            k_n, k_ms, k_w = eng.prof_collect()
            if not k_n:
                continue
            ms = k_ms / k_n
            nbytes = 8.0 * nk * nk + 8.0 * nk * Dk
            roof_ks[name] = dict(bound='hbm', kernel='kmatrix_kernel<%s> (full symmetric, n=%d, D=%d, ld=%d, 3 x 200 MB outputs in turn)' % (kind, nk, Dk, targets[0].stride(0)),
                                 achieved=nbytes / ms / 1e6, peak=HBM_PEAK_GBS, unit='GB/s', frac=nbytes / ms / 1e6 / HBM_PEAK_GBS,
                                 avg_launch_us=1e3 * ms, bytes_per_launch=nbytes, launches=k_n, traffic=None,
                                 note='the write-only ceiling of the same buffers is measured beside it (roofline_kmatrix_standalone.fill_); tile-shaped stores '
                                      'alone reach 0.66-0.72 of 8 TB/s at these sizes, 0.51-0.64 with rows off the 128-byte lines (profiles/r05_kmatrix_hbm_roofline.txt)')
        # a write-only fill_ of the same three buffers in turn, timed the same way (torch events on the engine's stream): the store
        # ceiling this kernel is compared with, measured in this run
        with eng.stream():
            st = torch.cuda.current_stream()
            for o in outs:
                o.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for r in range(30):
                outs[r % 3].fill_(1.0)
            e1.record(st)
        torch.cuda.synchronize()
        fill_ms = e0.elapsed_time(e1) / 30
        roof_ks['fill_'] = dict(what='torch fill_ of the same 200-MB buffers in turn (write-only ceiling, this run)', avg_launch_us=1e3 * fill_ms,
                                achieved=8.0 * nk * nk / fill_ms / 1e6, unit='GB/s', frac=8.0 * nk * nk / fill_ms / 1e6 / HBM_PEAK_GBS)
        del outs, outs_al

    ptab = vleg = sxleg = None
    if rank == 0 and args.prof_kernel != 'none':
        try:
            ptab = potrf_table(eng, torch, args.n)
        except Exception as exc:   # noqa: BLE001  (informational: must not cost the run its result line)
            ptab = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:200]))
        try:
            vleg = vecchia_leg(eng, torch)
        except Exception as exc:   # noqa: BLE001
            vleg = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:200]))
        try:
            sxleg = sexp_pair_leg(eng, torch)
        except Exception as exc:   # noqa: BLE001
            sxleg = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:200]))

    # ---- prediction leg: emulator with the imputations sharded over the ranks --------------------
    pred = None
    if not args.no_predict:
        from dgp_amd import emulator
        model.N = max(model.N, 1)
        for layer in model.all_layer:
            for nd in layer:
                nd.engine = eng
        est = model.estimate(burnin=0)
        # weak scaling like the training leg: every rank draws args.imputations imputations, the emulator's N is their total
        emu = emulator(est, N=args.imputations * world, seed=7, device=local)
        xt = np.random.default_rng(5).uniform(size=(args.predict_points, args.d))
        emu.predict(xt[:32])   # builds the per-imputation statistics + warms the kernels
        reps, tp = 0, 0.0
        while reps == 0 or tp < args.predict_seconds:   # (every rank takes the same number of passes: the time is the max over ranks)
            dd.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mu, var = emu.predict(xt)
            torch.cuda.synchronize()
            dd.barrier()
            tp += dd.allreduce_max_scalar(time.perf_counter() - t0, dev if world > 1 else None)
            reps += 1
        tp /= reps
        pred = dict(points=args.predict_points, imputations=args.imputations * world, imputations_per_rank=args.imputations, seconds=tp,
                    passes=reps, pts_per_s=args.predict_points / tp,
                    point_imputations_per_s=args.predict_points * args.imputations * world / tp, finite=bool(np.all(np.isfinite(mu)) and np.all(np.isfinite(var))))
        if args.prof_kernel != 'none':
            # the linked-GP pair kernel (the prediction leg's dominant kernel) against the f64 MFMA roofline, same call again
            # (by EVERY rank: predict ends in a collective; only rank 0 brackets its launches)
            if rank == 0:
                eng.prof_enable('linkgp_j')
            emu.predict(xt)
            p_n, p_ms, p_w = eng.prof_collect() if rank == 0 else (0, 0.0, 0.0)
            if p_n:
                ach = p_w / (p_ms * 1e-3) / 1e12
                pred['roofline_predict'] = dict(bound='mfma', kernel='linkgp_Jsep_kernel (Matern pair phase: record dot products on f64 MFMA)',
                                                achieved=ach, peak=F64_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / F64_PEAK_TFLOPS, traffic=None,
                                                launches=p_n, avg_launch_us=1e3 * p_ms / p_n, flop_convention='executed',
                                                J_elements_per_s=p_w / (args.d * 60.0) * 2.0 / (p_ms * 1e-3),
                                                J_elements_executed_per_s=p_w / (args.d * 60.0) / (p_ms * 1e-3),
                                                frac_of_f64_vector_peak=ach / F64_PEAK_TFLOPS,
                                                note='EXECUTED flops = M n^2/2 x D x 30 x 2 per launch: the record dot products of the lower pairs '
                                                     '(DESIGN section 3); SURVEY 8(d)\'s unit beside it: J elements per second, one element = one (i, j) of '
                                                     'J for one test point as the reference evaluates them (all n^2; the kernel evaluates the n^2/2 lower ones '
                                                     'and uses the symmetry), and the fraction of the f64 vector peak (= the matrix peak on gfx950)')
            # the first layer's predictor (functions.py:379-394: r^T R^-1 r on MFMA), a shorter pass of the same call
            xg = xt[:min(len(xt), 4096)]
            if rank == 0:
                eng.prof_enable('gp_quad')
            emu.predict(xg)
            g_n, g_ms, g_w = eng.prof_collect() if rank == 0 else (0, 0.0, 0.0)
            if g_n:
                ach = g_w / (g_ms * 1e-3) / 1e12
                pred['roofline_gp'] = dict(bound='mfma', kernel='gp_quad_kernel (r^T R^-1 r over the lower tiles of R^-1, f64 MFMA)', achieved=ach,
                                           peak=F64_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / F64_PEAK_TFLOPS, traffic=None, launches=g_n,
                                           avg_launch_us=1e3 * g_ms / g_n, points=len(xg), flop_convention='algorithmic / 2 = executed',
                                           note='algorithmic flops = 2 n^2 M per launch (R^-1 r, then the dot product, as the reference forms it); '
                                                'the kernel uses the symmetry of R^-1 and executes half of them')

    # ---- N > 1: ONE model trained by all ranks with the M-step's nodes split over them (dist.split_training; every rank
    #      runs the same I-step from the same seed, node i is fitted by rank i mod N, one all-gather per M-step) -- the
    #      informational strong-scaling companion of the replicas above
    split = None
    if world > 1 and not args.no_split_leg:
        # (informational: a failure here must not cost the run its result line)
        try:
            dd.split_training(nodes=True)
            shared, _, _ = build_model(args.n, args.d, 100, local)
            k_split = max(3, min(10, args.steps))
            for _ in range(2):
                shared.imp.sample(burnin=args.ess_burn)
                shared._m_step()
            dd.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k_split):
                shared.imp.sample(burnin=args.ess_burn)
                shared._m_step()
            torch.cuda.synchronize()
            dd.barrier()
            ts = dd.allreduce_max_scalar(time.perf_counter() - t0, dev)
            split = dict(what='one model, M-step nodes round-robin over the ranks (I-step replicated)', steps=k_split,
                         ms_per_step=1e3 * ts / k_split, si_it_per_s=k_split / ts)
        except Exception as exc:   # noqa: BLE001
            split = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:200]))
        finally:
            dd.split_training(nodes=False)

    # ---- N > 1: strong-scaling legs on the two axes north_star names (fixed total work, so the time should fall with N):
    #      cfg3-shaped prediction with the imputations sharded over the ranks (one all-reduce of the moments), and
    #      cfg4-shaped Vecchia training with the likelihood rows split (one all-reduce per speculative batch / optimiser round)
    strong = None
    if world > 1 and not args.no_strong_legs:
        strong = {}
        try:
            strong['cfg3_predict_imputations_sharded'] = strong_leg_cfg3(dd, torch, local, dev, world)
        except Exception as exc:   # noqa: BLE001  (informational: must not cost the run its result line)
            strong['cfg3_predict_imputations_sharded'] = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:200]))
        try:
            strong['cfg4_train_rows_split'] = strong_leg_cfg4(dd, torch, local, dev, world)
        except Exception as exc:   # noqa: BLE001
            strong['cfg4_train_rows_split'] = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:200]))
        finally:
            dd.split_training(rows=False, nodes=False)

    # the device stays busy for at least --min-gpu-seconds in all (untimed extra iterations; replicas: no collective inside)
    extra_steps = 0
    while time.perf_counter() - t_gpu0 < args.min_gpu_seconds:
        step()
        extra_steps += 1
    torch.cuda.synchronize()
    gpu_leg_seconds = time.perf_counter() - t_gpu0

    cpu = cpu_pred = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(model, counts, args.ess_burn)
        if pred is not None:
            try:
                cpu_pred = cpu_baseline_predict(model, args.d, args.imputations)
            except Exception as exc:   # noqa: BLE001  (informational: must not cost the run its result line)
                cpu_pred = dict(error='%s: %s' % (type(exc).__name__, str(exc)[:200]))

    dist_info = None
    if world > 1 or forced:   # RCCL's view of the job, in the record
        import torch.distributed as td
        ids = dd.allgather_vector(np.array([float(rank), float(local), float(torch.cuda.current_device())]), device=dev).reshape(-1, 3)
        dist_info = dict(world_size=td.get_world_size(), backend=td.get_backend(),
                         ranks=[dict(rank=int(r[0]), local_rank=int(r[1]), device='cuda:%d' % int(r[2])) for r in ids],
                         device_name=torch.cuda.get_device_name(local), devices_visible=torch.cuda.device_count(),
                         ranks_share_a_device=shared_device)

    if rank == 0:
        out = {
            'metric': 'SI iterations/sec (train) + predict pts/sec, 2-layer DGP n=2000 d=5',
            'value': value, 'unit': 'SI it/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * wall / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'scaling_metric': None if world == 1 else dict(
                columns=['predict.point_imputations_per_s', 'strong_scaling.cfg3_predict_imputations_sharded.point_imputations_per_s',
                         'strong_scaling.cfg4_train_rows_split.si_it_per_s', 'mstep_nodes_split.si_it_per_s'],
                note='`value` at N > 1 is N independent replica chains (the SI chain of one model does not shard): linear in N by construction and no evidence of '
                     'scaling.  north_star\'s "imputation-sample throughput to 8 GPUs" is answered by predict.point_imputations_per_s (weak: every rank draws its '
                     'imputations, one all-reduce of the moments) and by the strong_scaling legs (fixed total work over N ranks: the time should fall with N)'),
            'config': {'workload': 'configs[1]: 2-layer DGP, d=%d in / 1 out, n=%d, Matern-2.5, %d+1 GP nodes, '
                                   'train(ess_burn=%d): one step = imputer.sample() + the M-step = one iteration of dgp.train (dgp.py:1364-1412); '
                                   'the same rate over a train(N=%d) call is in `sustained`' % (args.d, args.n, args.d, args.ess_burn, args.sustained_steps),
                       'parallelism': ('replicas x%d (SI chain does not shard); predict: imputations sharded, 1 all-reduce' % world)
                       + ('' if world == 1 else '; `value` is replicas x rate: linear in N by construction -- the scaling figure of this run is '
                                                'strong_scaling.cfg3_predict_imputations_sharded.point_imputations_per_s (fixed total work)')},
            'step_split': step_split,
            'sustained': sustained, 'sustained_it_per_s': sustained['si_it_per_s'] if sustained else None,
            'predict': pred, 'counts': counts, 'roofline': roof, 'roofline_kmatrix': roof_k, 'roofline_kmatrix_standalone': roof_ks,
            'roofline_predict': (pred or {}).get('roofline_predict'), 'roofline_gp': (pred or {}).get('roofline_gp'),
            'roofline_predict_sexp': sxleg, 'roofline_nn': (vleg or {}).get('roofline_nn'),
            'potrf_table': ptab, 'vecchia': vleg, 'cpu_baseline': cpu, 'cpu_baseline_predict': cpu_pred, 'mstep_nodes_split': split,
            'speedup_vs_cpu_baseline': (value / world / cpu['value']) if cpu else None,
            'predict_speedup_vs_cpu_baseline': (pred['pts_per_s'] / cpu_pred['value']) if (pred and cpu_pred and cpu_pred.get('value')) else None,
            'strong_scaling': strong, 'distributed': dist_info,
            'gpu_legs': dict(seconds=gpu_leg_seconds, untimed_extra_steps=extra_steps),
        }
    if rank == 0:
        if c_stdio is not None:
            c_stdio.fflush(None)
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == '__main__':
    main()
