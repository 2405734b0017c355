"""Host logic that needs no GPU: the speculative-proposal bookkeeping of the device imputer equals
the sequential ESS loop (imputation.py:81-119), draw streams, sharding arithmetic, the 2-rank gloo
moment reduction."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import dgp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_speculative_angles_equal_sequential():
    from dgp_amd.imputation import shrink, speculative_angles, TWO_PI
    rng = np.random.default_rng(0)
    for _ in range(50):
        u = rng.random(12)
        ref = O.ess_angles(u)
        theta = TWO_PI * u[0]
        th, br = speculative_angles(theta, theta - TWO_PI, theta, u[1:])
        np.testing.assert_array_equal(np.array(th), ref)      # bit-exact: same expressions
        # continuing from the bracket after a fully rejected batch reproduces the tail
        th5, br5 = speculative_angles(theta, theta - TWO_PI, theta, u[1:5])
        t, lo, hi = shrink(th5[-1], *br5[-1], u[5])
        assert t == ref[5]


def sequential_reference(ll_fun, log_y, u):
    """imputation.py:85-119 with injected uniforms; returns (accepted index, uniforms consumed)."""
    th = O.ess_angles(u)
    for i, t in enumerate(th):
        if ll_fun(t) > log_y:
            return i, i + 1   # theta0's uniform + one per rejection
    raise RuntimeError


@pytest.mark.parametrize('B', [1, 2, 3, 8])
def test_batched_acceptance_consumes_like_sequential(B):
    """Emulate one_sample_block's control flow on the host with a synthetic log-likelihood."""
    from dgp_amd.imputation import DrawStream, shrink, speculative_angles, TWO_PI
    rng = np.random.default_rng(B)
    for trial in range(40):
        u = rng.random(40)
        ll = lambda t: -abs(np.sin(3 * t)) * 5.0 * (1 + trial % 3)   # ll(0) = 0 > log_y: the bracket always ends in acceptance
        log_y = -1.0
        idx, used = sequential_reference(ll, log_y, u)
        ds = DrawStream(z=[], u=list(u))
        theta = TWO_PI * ds.uniform_take(1)[0]
        lo, hi = theta - TWO_PI, theta
        consumed_before = 40 - len(ds._ubuf)
        found = None
        base = 0
        while found is None:
            us = ds.uniform_peek(B - 1)
            th, br = speculative_angles(theta, lo, hi, us)
            for b, t in enumerate(th):
                if ll(t) > log_y:
                    ds.uniform_take(b)
                    found = base + b
                    break
            else:
                ds.uniform_take(len(th) - 1)
                base += len(th)
                theta, (lo, hi) = th[-1], br[-1]
                theta, lo, hi = shrink(theta, lo, hi, ds.uniform_take(1)[0])
        assert found == idx
        assert 40 - len(ds._ubuf) == used


def test_drawstream_streams_are_independent_and_reproducible():
    from dgp_amd.imputation import DrawStream
    a, b = DrawStream(5), DrawStream(5)
    a.uniform_peek(7)                       # look-ahead must not disturb the normal stream
    np.testing.assert_array_equal(a.normal(4), b.normal(4))
    assert a.uniform_take(3) == b.uniform_take(3)
    c = DrawStream(6)
    assert c.uniform_take(1) != DrawStream(5).uniform_take(1)


def test_share_partitions_exactly():
    from dgp_amd.dist import share
    for total in (0, 1, 7, 10, 50):
        for w in (1, 2, 3, 8):
            parts = [share(total, r, w) for r in range(w)]
            assert sum(parts) == total and max(parts) - min(parts) <= 1
            from dgp_amd.dist import row_range
            rows = [row_range(total, r, w) for r in range(w)]
            assert rows[0][0] == 0 and rows[-1][1] == total and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))


def test_gloo_two_rank_moment_reduction(tmp_path):
    """world_size 2 over gloo: each rank accumulates its share of the reference's per-imputation
    predictions (golden g9); one all-reduce(sum); equals emulation.py:846-847 on all imputations."""
    script = tmp_path / 'worker.py'
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as td
from dgp_amd import dist as dd
dd.init_from_env('gloo')
assert dd.is_active() and dd.world() == 2
g = np.load(os.path.join(%r, 'tests', 'golden', 'g9_emulator_matern.npz'))
mu_s, var_s = g['mu_s'], g['var_s']
S = len(mu_s)
lo = sum(dd.share(S, r, 2) for r in range(dd.rank()))
cnt = dd.share(S, dd.rank(), 2)
s1 = torch.zeros(mu_s[0].shape, dtype=torch.float64); s2 = torch.zeros_like(s1)
for s in range(lo, lo + cnt):
    s1 += torch.from_numpy(mu_s[s]); s2 += torch.from_numpy(mu_s[s] ** 2 + var_s[s])
dd.allreduce_sum(s1, s2)
mu = s1 / S; var = s2 / S - mu ** 2
t = dd.allreduce_max_scalar(float(dd.rank()))
assert t == 1.0
# test points sharded instead (emulator(shard='points')): row blocks of unequal size gathered on every rank
for M in (7, 8, 1):
    full = np.arange(M * 3, dtype=float).reshape(M, 3)
    lo, hi = dd.row_range(M, dd.rank(), 2)
    got = dd.allgather_rows(full[lo:hi], M)
    assert got.shape == (M, 3) and np.array_equal(got, full)
np.testing.assert_allclose(mu.numpy(), g['mu'], rtol=1e-12, atol=1e-14)
np.testing.assert_allclose(var.numpy(), g['var'], rtol=1e-9, atol=1e-14)
# the two training splits (dist.split_training): Vecchia rows -> row blocks + sum of the (quad, logdet, gradient) vector;
# M-step nodes -> node i on rank i mod world + one equal-size all-gather of the fitted hyper-parameters.  They do not
# compose (ADVICE round 2): asking for both is an error and changes nothing.
assert dd.vecchia_rows(11) == (0, 11) and not dd.rows_split() and not dd.nodes_split()
try:
    dd.split_training(rows=True, nodes=True)
    raise SystemExit('rows + nodes accepted')
except ValueError:
    pass
assert not dd.rows_split() and not dd.nodes_split()
dd.split_training(rows=True)
try:
    dd.split_training(nodes=True)
    raise SystemExit('nodes accepted on top of rows')
except ValueError:
    pass
assert dd.rows_split() and not dd.nodes_split()
lo, hi = dd.vecchia_rows(11)
assert (lo, hi) == ((0, 6) if dd.rank() == 0 else (6, 11))
rows = np.arange(11 * 4, dtype=float).reshape(11, 4)
part = dd.allreduce_sum_vector(torch.from_numpy(rows[lo:hi].sum(0)))
assert np.array_equal(part.numpy(), rows.sum(0))
dd.split_training(rows=False, nodes=True)
assert dd.nodes_split() and dd.vecchia_rows(11) == (0, 11)
got = dd.allgather_vector(np.array([dd.rank(), 2.5, -1.0]))
assert got.shape == (2, 3) and np.array_equal(got[:, 0], [0.0, 1.0]) and np.all(got[:, 1] == 2.5)
# dgp._exchange_fits on stand-in nodes: 5 nodes over 2 ranks, every rank ends with every node's fit; a failure on ONE
# rank surfaces as the same LinAlgError on BOTH (so that train()'s restart happens everywhere), a lost hand-off as RuntimeError
import types
from dgp_amd.dgp import dgp as DGP
def mk(i, fitted):
    nd = types.SimpleNamespace(scale=np.array([1.0 + i if fitted else -1.0]), length=np.array([0.5 * i, 2.0] if i %% 2 else [3.0 + i]) * (1.0 if fitted else 0.0),
                               nugget=np.array([1e-6 * (i + 1) if fitted else 0.0]), path=[])
    nd.add_to_path = lambda nd=nd: nd.path.append((nd.scale[0], tuple(nd.length), nd.nugget[0]))
    return nd
me = types.SimpleNamespace(engine=None)
every = [(0, mk(i, i %% 2 == dd.rank())) for i in range(5)]
DGP._exchange_fits(me, every, None)
for i, (_, nd) in enumerate(every):
    ref = mk(i, True)
    assert nd.scale[0] == ref.scale[0] and np.array_equal(nd.length, ref.length) and nd.nugget[0] == ref.nugget[0], (i, nd)
    assert len(nd.path) == (0 if i %% 2 == dd.rank() else 1)
for exc, kind in ((np.linalg.LinAlgError('not PD'), np.linalg.LinAlgError), (RuntimeError('lost hand-off'), RuntimeError)):
    try:
        DGP._exchange_fits(me, every, exc if dd.rank() == 1 else None)
        raise SystemExit('no exception on rank ' + str(dd.rank()))
    except kind as e:
        assert 'rank(s) [1]' in str(e), str(e)
dd.split_training(rows=False, nodes=False)
dd.barrier()
print('rank', dd.rank(), 'ok')
''' % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', WORLD_SIZE='2')
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert 'ok' in o


def test_gloo_four_ranks_uneven_shares(tmp_path):
    """world_size 4 over gloo with remainders everywhere (verdict round 2, item 8): 50 imputations -> 13/13/12/12, the
    golden g9 imputations dealt 4 ways (some ranks get one more), row blocks whose last one is short or empty, the
    fixed-size all-gather of fitted hyper-parameters with 6 nodes on 4 ranks, and the N < world error path of
    emulator(shard=True) (raised before anything touches a device)."""
    script = tmp_path / 'worker4.py'
    script.write_text('''
import os, sys, types
sys.path.insert(0, %r)
import numpy as np, torch
from dgp_amd import dist as dd
dd.init_from_env('gloo')
W, r = dd.world(), dd.rank()
assert dd.is_active() and W == 4
assert [dd.share(50, q, W) for q in range(W)] == [13, 13, 12, 12]
g = np.load(os.path.join(%r, 'tests', 'golden', 'g9_emulator_matern.npz'))
mu_s, var_s = g['mu_s'], g['var_s']
S = len(mu_s)   # 3 imputations on 4 ranks: shares 1, 1, 1, 0 -- the rank without any still enters the collective
lo = sum(dd.share(S, q, W) for q in range(r)); cnt = dd.share(S, r, W)
s1 = torch.zeros(mu_s[0].shape, dtype=torch.float64); s2 = torch.zeros_like(s1)
for s in range(lo, lo + cnt):
    s1 += torch.from_numpy(mu_s[s]); s2 += torch.from_numpy(mu_s[s] ** 2 + var_s[s])
dd.allreduce_sum(s1, s2)
mu = s1 / S; var = s2 / S - mu ** 2
np.testing.assert_allclose(mu.numpy(), g['mu'], rtol=1e-12, atol=1e-14)
np.testing.assert_allclose(var.numpy(), g['var'], rtol=1e-9, atol=1e-14)
for M in (50, 9, 5, 2):      # 9 -> blocks 3,3,3,0; 5 -> 2,2,1,0; 2 -> 1,1,0,0
    full = np.arange(M * 2, dtype=float).reshape(M, 2)
    a, b = dd.row_range(M, r, W)
    got = dd.allgather_rows(full[a:b], M)
    assert got.shape == (M, 2) and np.array_equal(got, full), (M, r)
assert dd.allreduce_max_scalar(float(r)) == 3.0
# node split: 6 nodes on 4 ranks (2, 2, 1, 1), one all-gather of doubles
from dgp_amd.dgp import dgp as DGP
def mk(i, fitted):
    nd = types.SimpleNamespace(scale=np.array([2.0 + i if fitted else 0.0]), length=np.array([0.1 * (i + 1)] * (1 + i %% 3)) * (1.0 if fitted else 0.0),
                               nugget=np.array([1e-8 * (i + 1) if fitted else 0.0]), path=[])
    nd.add_to_path = lambda nd=nd: nd.path.append(1)
    return nd
dd.split_training(nodes=True)
every = [(0, mk(i, i %% W == r)) for i in range(6)]
DGP._exchange_fits(types.SimpleNamespace(engine=None), every, None)
for i, (_, nd) in enumerate(every):
    ref = mk(i, True)
    assert nd.scale[0] == ref.scale[0] and np.array_equal(nd.length, ref.length) and nd.nugget[0] == ref.nugget[0], (r, i)
dd.split_training(nodes=False)
# N < world: every rank raises before building an engine (no collective is entered by anyone)
from dgp_amd.emulation import emulator
try:
    emulator([[types.SimpleNamespace(vecch=False, type='gp')]], N=3, shard=True)
    raise SystemExit('N < world accepted')
except Exception as e:
    assert 'at least one imputation per rank' in str(e), str(e)
dd.barrier()
print('rank', r, 'ok')
''' % (ROOT, ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', WORLD_SIZE='4')
    procs = []
    for r in range(4):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert 'ok' in o


def test_lockstep_lbfgsb_matches_scipy_minimize():
    """dgp_amd.mstep.minimize_lockstep drives scipy's L-BFGS-B core for several problems at once; every problem
    must see exactly the iterates scipy.optimize.minimize(method='L-BFGS-B', jac=True) produces on its own."""
    from scipy.optimize import minimize, Bounds
    from dgp_amd import mstep
    if not mstep._HAVE_CORE:
        pytest.skip('scipy L-BFGS-B core differs from the one the driver was written for')

    def rosen(x):
        f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
        g = np.zeros_like(x)
        g[:-1] = -400.0 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
        return np.atleast_1d(f), g

    def quartic(x):
        return np.atleast_1d(np.sum((x - 0.3) ** 4 + 0.5 * x ** 2)), 4 * (x - 0.3) ** 3 + x

    def logistic(x):
        e = np.exp(-x)
        return np.atleast_1d(np.sum(np.log1p(e) + 0.1 * x ** 2)), -e / (1 + e) + 0.2 * x

    cases = [(rosen, np.array([-1.2, 1.0, 0.5]), None, None, dict(maxiter=100, maxfun=35)),
             (quartic, np.array([2.0, -1.0]), np.array([-np.inf, -0.5]), np.array([1.5, np.inf]), dict(maxiter=100, maxfun=30)),
             (logistic, np.array([3.0, -2.0, 0.1, 0.7]), np.array([-1.0, -np.inf, -np.inf, 0.2]),
              np.array([np.inf, np.inf, 13.0, 0.9]), dict(maxiter=3, maxfun=30)),
             (rosen, np.array([0.0, 0.0]), None, None, dict(maxiter=100, maxfun=5))]
    ref_pts, ref_last = [], []
    for fun, x0, lb, ub, opts in cases:
        pts = []

        def logged(x, fun=fun, pts=pts):
            pts.append(np.array(x, dtype=float))
            return fun(x)
        kw = {} if lb is None else dict(bounds=Bounds(lb, ub))
        res = minimize(logged, x0, method='L-BFGS-B', jac=True, options=opts, **kw)
        ref_pts.append(pts)
        ref_last.append(res.x)
    got = [[] for _ in cases]

    def evaluate(req):
        out = []
        for i, x in req:
            got[i].append(x.copy())
            out.append(cases[i][0](x))
        return out
    probs = [mstep._Problem(x0, lb, ub, o['maxiter'], o['maxfun']) for _, x0, lb, ub, o in cases]
    rounds = mstep.minimize_lockstep(probs, evaluate)
    assert rounds == max(len(p) for p in ref_pts)
    for i in range(len(cases)):
        assert len(got[i]) == len(ref_pts[i]), (i, len(got[i]), len(ref_pts[i]))
        for a, b in zip(got[i], ref_pts[i]):
            assert np.array_equal(a, b)
        assert np.array_equal(probs[i].x, ref_last[i])
    # several groups in flight (launch / collect halves, the Vecchia M-step's driver): another launch order, the same iterates;
    # a group's results are collected only after the OTHER groups' launches have been queued behind it
    for groups in (2, 3):
        got2 = [[] for _ in cases]
        log, hooks = [], []

        def launch(req, slot):
            log.append(('launch', slot, tuple(i for i, _ in req)))
            for i, x in req:
                got2[i].append(x.copy())
            return [cases[i][0](x) for i, x in req]

        def collect(req, token):
            log.append(('collect', tuple(i for i, _ in req)))
            return token

        def evaluate2(req):
            return collect(req, launch(req, 0))
        evaluate2.launch, evaluate2.collect, evaluate2.abandon = launch, collect, lambda token: None
        probs2 = [mstep._Problem(x0, lb, ub, o['maxiter'], o['maxfun']) for _, x0, lb, ub, o in cases]
        rounds2 = mstep.minimize_lockstep(probs2, evaluate2, after_first_launch=lambda: hooks.append(len(log)), groups=groups)
        assert hooks == [groups] and all(e[0] == 'launch' for e in log[:groups])   # the hook runs once, behind every group's first launch
        assert rounds2 == max(len(p) for p in ref_pts)
        for i in range(len(cases)):
            assert len(got2[i]) == len(ref_pts[i])
            assert all(np.array_equal(a, b) for a, b in zip(got2[i], ref_pts[i]))
            assert np.array_equal(probs2[i].x, ref_last[i])
        slots = {e[1] for e in log if e[0] == 'launch'}
        assert slots == set(range(groups))


def test_drawstream_prefetch_keeps_the_sequence():
    """Normals pre-generated on the background thread are the ones the generator would have produced in place."""
    from dgp_amd.imputation import DrawStream
    a, b = DrawStream(123), DrawStream(123)
    ref = np.concatenate([a.normal(7), a.normals(50), a.normal(13), a.normals(100), a.normal(5)])
    out = [b.normal(7)]
    b.prefetch(40)            # fewer than the next request: topped up in place
    out.append(b.normals(50))
    b.prefetch(200)           # more than needed: the rest is kept for later requests
    out.append(b.normal(13))
    out.append(b.normals(100))
    b.prefetch(3)
    out.append(b.normal(5))
    assert np.array_equal(np.concatenate(out), ref)
    import pickle
    b.prefetch(10)
    c = pickle.loads(pickle.dumps(b))
    assert np.array_equal(c.normals(10), a.normals(10))


def test_structure_exported_from_dgpsi_loads(golden, tmp_path):
    """tools/export_dgpsi_structure.py run on a structure trained by the reference (oracle/gen_golden.py:gen_export)
    -> dgp_amd.load_structure: same hyper-parameters, latents, wiring; and a Hetero likelihood node survives
    save_structure / load_structure."""
    from conftest import case, GOLDEN
    from dgp_amd.utils import load_structure, save_structure
    from dgp_amd import kernel, Hetero
    chk = golden('g16_dgpsi_export_check')
    layers = load_structure(os.path.join(GOLDEN, 'g16_dgpsi_export_structure'))
    assert len(layers) == int(chk['est_n_layer'])
    for l, layer in enumerate(layers):
        assert len(layer) == int(chk['est_l%d_n' % l])
        for k, nd in enumerate(layer):
            c = case(chk, 'est_l%d_k%d_' % (l, k))
            assert nd.name == str(c['name']) and nd.scale_est == bool(c['scale_est']) and nd.nugget_est == bool(c['nugget_est'])
            for a in ('length', 'scale', 'nugget', 'input', 'output', 'input_dim'):
                assert np.array_equal(getattr(nd, a), c[a]), a
            assert (nd.global_input is not None) == bool(c['has_global'])
            if nd.global_input is not None:
                assert np.array_equal(nd.global_input, c['global_input']) and np.array_equal(nd.connect, c['connect'])
            assert nd.D == nd.input.shape[1] + (0 if nd.global_input is None else nd.global_input.shape[1])
    rng = np.random.default_rng(0)
    g1, g2, h = kernel(length=np.array([0.7])), kernel(length=np.array([1.2]), name='matern2.5'), Hetero()
    for g in (g1, g2):
        g.input, g.output, g.input_dim = rng.uniform(size=(6, 1)), rng.normal(size=(6, 1)), np.array([0])
        g.para_path = np.atleast_2d(np.concatenate((g.scale, g.length, g.nugget)))
    h.rep = np.array([0, 0, 1, 2, 3, 4, 5, 5])
    h.input, h.output, h.input_dim = rng.normal(size=(8, 2)), rng.normal(size=(8, 1)), np.array([0, 1])
    save_structure([[g1, g2], [h]], str(tmp_path / 'het'))
    back = load_structure(str(tmp_path / 'het'))
    assert back[1][0].name == 'Hetero' and back[1][0].type == 'likelihood' and np.array_equal(back[1][0].rep, h.rep)
    assert np.array_equal(back[1][0].input, h.input) and np.array_equal(back[1][0].output, h.output)
    assert back[0][1].name == 'matern2.5' and np.array_equal(back[0][1].output, g2.output)


def test_count_likelihood_nodes_match_reference(golden):
    """dgp_amd.Poisson / NegBin (host plugin protocol) and dgp's latent warm starts for them against the reference's
    values (g17_count_likelihoods)."""
    from dgp_amd import Poisson, NegBin, ZIP, ZINB
    from dgp_amd.likelihood_class import ghdiag
    from dgp_amd.dgp import dgp
    g = golden('g17_count_likelihoods')
    for name, cls in (('poisson', Poisson), ('negbin', NegBin), ('zip', ZIP), ('zinb', ZINB)):
        h = cls()
        h.input, h.output = g[name + '_input'], g[name + '_output']
        assert h.type == 'likelihood' and h.exact_post_idx is None and h.rep is None
        np.testing.assert_allclose(h.llik(), float(g[name + '_llik']), rtol=1e-12)
        pm, pv = h.prediction(g[name + '_m'], g[name + '_v'])
        np.testing.assert_allclose(pm, g[name + '_pm'], rtol=1e-13)
        np.testing.assert_allclose(pv, g[name + '_pv'], rtol=1e-13)
        np.testing.assert_allclose(ghdiag(h.pllik, g[name + '_m'], g[name + '_v'], g[name + '_yq']), g[name + '_gh'], rtol=1e-12)
        q = {'poisson': 1, 'negbin': 2, 'zip': 2, 'zinb': 3}[name]
        assert h.sampling(np.zeros((5, q))).shape == (5,)
        for tag in ('norep', 'rep'):
            pre = 'ws_%s_%s_' % (name, tag)
            X, Y = g[pre + 'X'], g[pre + 'Y']
            obj = dgp.__new__(dgp)
            obj.Y, obj.indices = Y, None
            X0, inv = np.unique(X, return_inverse=True, axis=0)
            obj.X = X0 if len(X0) != len(X) else X
            if len(X0) != len(X):
                obj.indices = np.asarray(inv).reshape(-1)
            obj.all_layer, obj.n_layer = [[None] * q, [cls()]], 2
            lat = obj._count_warm_start(0)
            cols = [0] if (name == 'negbin' and tag == 'norep') else list(range(q))
            np.testing.assert_allclose(lat[:, cols], g[pre + 'latent'][:, cols], rtol=1e-13)
            assert np.all(np.isfinite(lat))


def test_categorical_likelihood_matches_reference(golden):
    """dgp_amd.Categorical (logit / probit / softmax / robustmax) and dgp's latent warm starts for it against the
    reference's values (g18_categorical); the Monte-Carlo predictions use numpy's global stream like the reference."""
    from dgp_amd import Categorical
    from dgp_amd.dgp import dgp
    g = golden('g18_categorical')
    for tag, K in (('logit', 2), ('probit', 2), ('softmax', 3), ('robustmax', 4)):
        h = Categorical(num_classes=K, link=tag)
        h.input = g[tag + '_input']
        h.output = g[tag + '_output'] if K == 2 else g[tag + '_output'].astype(int)
        np.testing.assert_allclose(h.llik(), float(g[tag + '_llik']), rtol=1e-12)
        np.random.seed(99)
        pm, pv = h.prediction(g[tag + '_m'], g[tag + '_v'])
        np.testing.assert_allclose(pm, g[tag + '_pm'], rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(pv, g[tag + '_pv'], rtol=1e-10, atol=1e-15)
        yq = g[tag + '_yq'] if K == 2 else g[tag + '_yq'].astype(int)
        np.testing.assert_allclose(h.pllik(yq.reshape(-1, 1, 1), g[tag + '_fq']), g[tag + '_pllik'], rtol=1e-12)
        np.testing.assert_allclose(h.sampling(h.input), g[tag + '_samp'], rtol=1e-13)
    for tag, K in (('bin', 2), ('multi', 3)):
        for rtag in ('norep', 'rep'):
            pre = 'ws_%s_%s_' % (tag, rtag)
            X = g[pre + 'X']
            obj = dgp.__new__(dgp)
            obj.Y, obj.indices = g[pre + 'Ycode'].astype(int), None
            X0, inv = np.unique(X, return_inverse=True, axis=0)
            obj.X = X0 if len(X0) != len(X) else X
            if len(X0) != len(X):
                obj.indices = np.asarray(inv).reshape(-1)
            obj.n_data = len(obj.X)
            lik = Categorical(num_classes=K, link=str(g[pre + 'link']))
            assert int(g[pre + 'K']) == K
            obj.all_layer, obj.n_layer = [[None] * (1 if K == 2 else K), [lik]], 2
            np.testing.assert_allclose(obj._categorical_warm_start(0), g[pre + 'latent'], rtol=1e-13)


def test_public_api_accepts_the_reference_calls():
    """Every public method of the classes dgpsi exports (names and parameter names read from the reference's sources by
    oracle/gen_api_signatures.py -> tests/golden/api_signatures.json) exists here and accepts the same parameters.
    The only absences are numerical helpers whose work moved into device kernels."""
    import inspect
    import json
    import dgp_amd
    api = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'api_signatures.json')))
    moved_to_device = {('Hetero', 'post_het1'), ('Hetero', 'post_het2'), ('Hetero', 'post_het_vecch'),
                       ('Hetero', 'posterior_vecch'),                    # -> Engine.post_het / vecchia_post_het
                       ('emulator', 'predict_mice'), ('emulator', 'predict_mice_2layer_likelihood'),
                       ('emulator', 'predict_vigf'), ('emulator', 'predict_vigf_2layer_likelihood')}   # -> the layer walk
    missing, bad = set(), []
    for cname, meths in api['classes'].items():
        cls = getattr(dgp_amd, cname)
        for m, ps in meths.items():
            f = getattr(cls, m, None)
            if f is None:
                missing.add((cname, m))
                continue
            sig = inspect.signature(f).parameters
            if any(p.kind == p.VAR_KEYWORD for p in sig.values()):
                continue
            bad += [(cname, m, p) for p in ps if p not in sig]
    for fn, ps in api['functions'].items():
        sig = inspect.signature(getattr(dgp_amd, fn)).parameters
        bad += [(fn, p) for p in ps if p not in sig]
    assert missing == moved_to_device, missing ^ moved_to_device
    assert not bad, bad


def test_operator_api_shims_keep_the_reference_argument_lists():
    """dgp_amd.functions / dgp_amd.vecchia (the njit operator API of SURVEY.md 8(b)): every function the reference's
    callers import exists under the same name and takes the reference's parameters in the reference's ORDER (read from its
    sources by oracle/gen_api_signatures.py), so positional calls keep working; the only addition is a trailing
    keyword `engine`.  Parsed with ast: importing the modules needs a HIP device."""
    import ast
    import json
    api = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'api_signatures.json')))['operators']
    assert set(api) == {'functions', 'vecchia'}
    for mod, funcs in api.items():
        tree = ast.parse(open(os.path.join(ROOT, 'dgp_amd', mod + '.py')).read())
        have = {n.name: [a.arg for a in n.args.args] for n in tree.body if isinstance(n, ast.FunctionDef)}
        for fn, ps in funcs.items():
            assert fn in have, (mod, fn)
            assert have[fn][:len(ps)] == ps and have[fn][len(ps):] == ['engine'], (mod, fn, have[fn], ps)


def test_lost_handoff_is_not_a_numerical_failure():
    """info < 0 (a bounded in-kernel spin gave up) raises DgpAmdError, which dgp.train's LinAlgError restart policy
    (dgp.py:1402-1412) does not swallow; info > 0 stays numpy's LinAlgError."""
    import numpy as np
    import pytest
    from dgp_amd.ops import raise_not_pd, DgpAmdError
    with pytest.raises(np.linalg.LinAlgError):
        raise_not_pd(17)
    with pytest.raises(DgpAmdError):
        raise_not_pd(-1)
    assert not issubclass(DgpAmdError, np.linalg.LinAlgError)


def test_llik_finish_scalar_path_is_bit_identical():
    """kernel._llik_finish has a python-float path for the common case (no replicates, gamma / inverse-gamma or no prior, one
    or two parameters) and a lean numpy path for several lengthscales; both must reproduce the general numpy path bit for bit -- both feed L-BFGS-B, whose iterates the
    lock-step M-step promises to leave unchanged."""
    import inspect
    import textwrap
    import types
    from dgp_amd import kernel_class as kc
    src = textwrap.dedent(inspect.getsource(kc.kernel._llik_finish))
    general = src.replace("if self.rep is None and self.prior_name in (None, 'ga', 'inv_ga') and P <= 2:", "if False:")
    assert general != src
    lean = "if self.rep is None and self.prior_name in (None, 'ga', 'inv_ga'):"   # the several-lengthscale path without numpy's wrappers
    assert lean in general
    general = general.replace(lean, "if False:")
    ns = {}
    exec("import numpy as np, math\n" + general, ns)
    rng = np.random.default_rng(3)

    class Node:
        pass
    for trial in range(600):
        a = Node()
        a.output = np.zeros((int(rng.integers(5, 3000)), 1))
        a.rep = None
        a.prior_name = [None, 'ga', 'inv_ga'][trial % 3]
        a.prior_coef = rng.uniform(0.5, 3, size=2)
        a.scale_est = bool(trial % 2)
        a.scale = np.array([rng.uniform(0.1, 5)])
        a.nugget_est = bool((trial // 2) % 2)
        a.length = rng.uniform(0.1, 5, size=1 if trial % 5 < 2 else int(rng.integers(2, 12)))   # (one shared or one per input dimension)
        a.nugget = np.array([rng.uniform(1e-8, 1e-2)])
        a._raise_if_not_pd = lambda v: None
        b = Node()
        b.__dict__.update({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in a.__dict__.items()})
        for o in (a, b):
            for nm in ('log_prior', 'log_prior_fod', 'gfod'):
                setattr(o, nm, types.MethodType(getattr(kc.kernel, nm), o))
        P = len(a.length) + int(a.nugget_est)
        host = np.concatenate(([rng.normal() * 100, rng.uniform(1, 5000)], rng.normal(size=2 * P) * 50, [0.0]))
        fa, ga = kc.kernel._llik_finish(a, host)
        fb, gb = ns['_llik_finish'](b, host)
        assert np.array_equal(np.ravel(fa), np.ravel(fb)) and np.array_equal(np.ravel(ga), np.ravel(gb))
        assert np.array_equal(np.ravel(a.scale), np.ravel(b.scale))


def test_cell_order_groups_rows_into_separated_cells():
    """ops.cell_order (the order the emulator hands a linked Matern node's training points to the pair kernel in): a
    permutation; every part produced by a split is aligned to 64 rows while larger than a block and to 16 below; two parts
    separated by a split in coordinate k satisfy max <= min in that coordinate (what linkgp_Jsep_kernel's class bounds test);
    with D coordinates in turn about half of the (16-row cell, 64-row block, coordinate) triples are separated at n = 2000, D = 5."""
    import importlib
    ops_src = open(os.path.join(ROOT, 'dgp_amd', 'ops.py')).read()
    ns = {'np': np}
    start = ops_src.index('def cell_order(')
    end = ops_src.index('\n\n\n', start)
    exec(ops_src[start:end], ns)   # (the function alone: importing dgp_amd.ops needs the HIP library)
    cell_order = ns['cell_order']
    rng = np.random.default_rng(3)
    for n, D in [(2000, 5), (1990, 5), (70, 2), (16, 3), (17, 3), (130, 1), (5000, 10)]:
        W = rng.normal(size=(n, D))
        W[: n // 4, 0] = np.round(W[: n // 4, 0], 1)   # ties
        p = cell_order(W)
        assert sorted(p.tolist()) == list(range(n))
        Wp = W[p]
        if n > 64:   # the first split is in coordinate 0, at a multiple of 64 rows
            left = 64 * max(1, int(round(n / 128.0)))
            assert Wp[:left, 0].max() <= Wp[left:, 0].min()
        if (n, D) == (2000, 5):
            nb, nc = (n + 63) // 64, (n + 15) // 16
            blo = np.array([Wp[64 * b:64 * b + 64].min(0) for b in range(nb)]); bhi = np.array([Wp[64 * b:64 * b + 64].max(0) for b in range(nb)])
            clo = np.array([Wp[16 * c:16 * c + 16].min(0) for c in range(nc)]); chi = np.array([Wp[16 * c:16 * c + 16].max(0) for c in range(nc)])
            sep = (chi[:, None, :] <= blo[None, :, :]) | (clo[:, None, :] > bhi[None, :, :])
            lower = (np.arange(nc)[:, None] // 4) > np.arange(nb)[None, :]   # cells of block rows strictly below the diagonal
            frac = sep[lower].mean()
            assert 0.42 < frac < 0.6, frac
