"""The HIP path at BASELINE.json's full sizes: size-independent properties where the oracle is too slow (cfg3's n = 5000
factorisation and inverse, cfg2's n = 2000 predictors, cfg4's n = 50 000 Vecchia kernels) and direct comparisons with the
oracle where it finishes in seconds -- its per-test-point predictors dealt to host processes (tests/oracle_pool.py): the
Matern link_gp of cfg2's output node at n = 2000 with 5 + 5 inputs and emulator.predict of the bench model against the
oracle's layer walk (round 6), cfg3's SExp link_gp and block update at n = 5000, cfg5's chain at n = 1000.
Needs an MI355X: -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    from dgp_amd.ops import Engine
    return Engine(0)


def close(a, b, rtol=1e-10, atol=1e-13):
    np.testing.assert_allclose(np.asarray(a, float), np.asarray(b, float), rtol=rtol, atol=atol)


def npy(t):
    return t.detach().cpu().numpy()


def test_cfg3_factorisation_round_trip(eng):
    """n = 5000, 10 SExp inputs (cfg3): log-determinant against LAPACK, K (K^-1 v) = v and K (K^-1 y) = y for the
    right-hand side that rides along, symmetry of the inverse."""
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(50)
    n, D, nug = 5000, 10, 1e-3
    X = rng.uniform(size=(n, D))
    y = rng.normal(size=n)
    length = np.full(D, 1.2)
    K = O.k_matrix(X, length, nug, 'sexp')
    A = eng.kmatrix('sexp', eng.tensor(X), None, None, length, nug, full=False, Y=eng.tensor(y))
    logdet, info = eng.potrf(n, A)
    work = eng.potrf_workspace(n, 1)
    Np = eng.padded_dim(n)
    Ainv = eng.empty(Np, Np)
    eng.potri(n, A, Ainv, 1, work)
    eng.sync()
    assert int(npy(info)[0]) == 0
    sign, ld_ref = np.linalg.slogdet(K)
    assert sign > 0
    close(npy(logdet)[0], ld_ref, rtol=1e-9)
    Ai = npy(Ainv)
    Kinv, alpha = Ai[:n, :n], -Ai[n, :n]
    assert np.array_equal(Kinv, Kinv.T)
    V = rng.normal(size=(n, 3))
    R = K @ (Kinv @ V) - V
    assert np.abs(R).max() < 1e-8 * np.abs(Kinv @ V).max()
    assert np.abs(K @ alpha - y).max() < 1e-8 * np.abs(alpha).max()


@pytest.mark.parametrize('name', ['sexp', 'matern2.5'])
def test_cfg2_predictors_at_full_size(eng, name):
    """n = 2000 (cfg2): (1) the GP predictor at its own training points: mean_i = y_i - eta (R^-1 y)_i and
    var_i = scale (2 eta - eta^2 (R^-1)_ii); (2) the linked-GP predictor with zero input variance is the GP predictor
    (MFMA SExp kernel / separable Matern kernel, 256 test points, with global inputs); (3) it is linear in R^-1 y."""
    rng = np.random.default_rng(20)
    n, Dw, Dz, eta, scale = 2000, 5, 5, 1e-4, 1.3
    X = rng.uniform(size=(n, Dw + Dz))
    y = rng.normal(size=n)
    length = rng.uniform(0.8, 1.6, size=Dw + Dz)
    Xd = eng.tensor(X)
    A = eng.kmatrix(name, Xd, None, None, length, eta, full=False, Y=eng.tensor(y))
    _, info = eng.potrf(n, A)
    work = eng.potrf_workspace(n, 1)
    Np = eng.padded_dim(n)
    Ainv = eng.empty(Np, Np)
    eng.potri(n, A, Ainv, 1, work)
    assert int(npy(info)[0]) == 0
    ry = (-Ainv[n, :n]).contiguous()
    m, v = eng.gp_predict(name, Xd, Xd, length, Ainv, Np, ry, scale, eta)
    ryh, dg = npy(ry), np.diag(npy(Ainv)[:n, :n])
    close(npy(m), y - eta * ryh, rtol=1e-7, atol=1e-8)
    close(npy(v), scale * (2 * eta - eta ** 2 * dg), rtol=1e-4, atol=1e-9)
    M = 256
    xt = rng.uniform(size=(M, Dw + Dz))
    xtd = eng.tensor(xt)
    gm, gv = eng.gp_predict(name, xtd, Xd, length, Ainv, Np, ry, scale, eta)
    args = (eng.tensor(xt[:, :Dw]), eng.zeros(M, Dw), eng.tensor(xt[:, Dw:]), eng.tensor(X[:, :Dw]), eng.tensor(X[:, Dw:]),
            length, Ainv, Np)
    lm, lv = eng.linkgp_predict(name, *args, ry, scale, eta)
    close(npy(lm), npy(gm), rtol=1e-8, atol=1e-9)
    close(npy(lv), npy(gv), rtol=1e-5, atol=1e-7 * scale)
    lm3, _ = eng.linkgp_predict(name, *args, (3.0 * ry).contiguous(), scale, eta)
    close(npy(lm3), 3.0 * npy(lm), rtol=1e-8, atol=1e-9)      # (sums of O(1e4) terms: rounding of 3 ry, not of the kernel)
    # a genuinely uncertain input can only widen the prediction on average and shrinks the mean towards zero
    vv = eng.tensor(np.full((M, Dw), 0.05))
    um, uv = eng.linkgp_predict(name, args[0], vv, *args[2:], ry, scale, eta)
    assert np.all(npy(uv) > 0) and np.mean(npy(uv)) > np.mean(npy(lv))


def test_cfg4_vecchia_at_full_size(eng):
    """n = 50 000, d = 8, m = 25 (cfg4): ordered neighbour rows against brute force (bit-exact), the sparse forward
    solve through its defining recurrence on all rows, the likelihood terms on a self-contained prefix against the
    oracle and through their scaling laws on the full set."""
    import torch
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(40)
    n, d, m = 50000, 8, 25
    X = rng.uniform(size=(n, d))
    y = np.sin(X @ rng.normal(size=d)) + 0.1 * rng.normal(size=n)
    length = np.full(d, 0.7)
    Xs = X / length
    NNd = eng.nn_ordered(eng.tensor(Xs), m)
    NN = npy(NNd)
    assert NN.shape == (n, m + 1) and np.array_equal(NN[:, 0], np.arange(n))
    for i in [0, 1, 5, 25, 26, 51, 52, 4095, 4096, 12345, 33333, n - 1]:
        di = ((Xs[:i + 1] - Xs[i]) ** 2).sum(1)
        want = np.sort(np.lexsort((np.arange(i + 1), di))[:m + 1])[::-1]
        assert np.array_equal(NN[i][:len(want)], want) and np.all(NN[i][len(want):] == -1)
    Xd, yd, ones = eng.tensor(X), eng.tensor(y), eng.tensor(np.ones(n))
    Lm = eng.vecchia_lmatrix('matern2.5', Xd, NNd, length, 1e-4)
    b = rng.normal(size=n)
    x = npy(eng.vecchia_spsolve(Lm, NNd, 1.0, eng.tensor(b)))
    L = npy(Lm)
    idx = np.where(NN >= 0, NN, 0)
    resid = (np.where(NN >= 0, L, 0.0) * x[idx]).sum(1) - b
    assert np.abs(resid).max() < 1e-9 * max(1.0, np.abs(x).max())
    out = npy(eng.vecchia_llik('matern2.5', Xd, yd, NNd, length, 1e-4, ones))
    out2 = npy(eng.vecchia_llik('matern2.5', Xd, eng.tensor(2.0 * y), NNd, length, 1e-4, ones))
    close(out2[0], 4.0 * out[0], rtol=1e-12)      # quadratic form scales with y^2
    close(out2[1], out[1], rtol=0, atol=0)        # log-determinant does not see y
    k = 1500                                      # rows 0..k-1 only point at earlier rows: a self-contained problem
    sub = npy(eng.vecchia_llik('matern2.5', eng.tensor(X[:k]), eng.tensor(y[:k]), eng.tensor(NN[:k], dtype=torch.int64),
                               length, 1e-4, eng.tensor(np.ones(k))))
    ref = O.vecchia_llik(X[:k], y[:k], NN[:k], 1.0, length, 1e-4, np.ones(k), 'matern2.5')
    close(-0.5 * (sub[1] + sub[0]), ref, rtol=1e-9)
    assert sub[0] < out[0] and np.isfinite(out).all()
    # LATE rows, through the full-size arrays (a defect confined to late row blocks / high tile indices would
    # pass the prefix): the kernels take a block of rows of the neighbour array and index X, y through its entries, so three
    # windows -- the middle, the 31st-40th thousand, the very end -- are evaluated on the device from the WHOLE arrays and on
    # the host by the oracle's per-row arithmetic (vecchia.py:164-180, 182-242: the same statements as oracle.vecchia_llik /
    # vecchia_nllik, summed over the window's rows only).  Likelihood sums at 1e-10, gradient sums at 1e-8.
    from scipy.linalg import solve_triangular
    for lo, hi in ((20000, 21500), (38750, 39250), (48500, 50000)):
        rows = NN[lo:hi]
        quad = logdet = 0.0
        dq, dl = np.zeros(d), np.zeros(d)
        for row in rows:
            idx = row[row >= 0][::-1]
            bsz = len(idx)
            Ki, dKi = O.k_matrix_fod(X[idx], length, 0.0, 'matern2.5', False)
            Ki[np.arange(bsz), np.arange(bsz)] = 1.0 + 1e-4
            Li = np.linalg.cholesky(Ki)
            w = solve_triangular(Li, y[idx], lower=True)
            e = np.zeros(bsz)
            e[-1] = 1.0
            u = solve_triangular(Li.T, e, lower=False)
            for k in range(d):
                tk = solve_triangular(Li, dKi[k] @ u, lower=True)
                dq[k] += 2 * (w @ tk) * w[-1] - tk[-1] * w[-1] ** 2
                dl[k] += tk[-1]
            quad += w[-1] ** 2
            logdet += 2 * np.log(Li[-1, -1])
        rd = eng.tensor(rows, dtype=torch.int64)
        got = npy(eng.vecchia_llik('matern2.5', Xd, yd, rd, length, 1e-4, ones))
        close(got, [quad, logdet], rtol=1e-10, atol=1e-9)
        gn, P = eng.vecchia_nllik('matern2.5', Xd, yd, rd, length, 1e-4, ones, False)
        gn = npy(gn)
        assert P == d
        close(gn[:2], [quad, logdet], rtol=1e-10, atol=1e-9)
        close(gn[2:2 + d], dq, rtol=1e-8, atol=1e-8)
        close(gn[2 + d:2 + 2 * d], dl, rtol=1e-8, atol=1e-8)


def _node(eng, name, length, inp, out, glob=None, scale=1.0, scale_est=False, nugget=1e-6):
    """A dgp_amd kernel node with its data attached (what dgp() would build)."""
    from dgp_amd import kernel
    nd = kernel(length=np.asarray(length, float).copy(), scale=scale, nugget=nugget, name=name, scale_est=scale_est,
                input_dim=np.arange(inp.shape[1]), connect=None if glob is None else np.arange(glob.shape[1]), engine=eng)
    nd.input, nd.output, nd.global_input = inp.copy(), out.reshape(-1, 1).copy(), None if glob is None else glob.copy()
    nd.vecch = False
    nd.D = inp.shape[1] + (0 if glob is None else glob.shape[1])
    return nd


def test_cfg3_block_update_at_full_size_vs_oracle(eng):
    """BASELINE configs[2] at its own size (n = 5000, d = 10 in / 3 out, SExp): ONE block update of the hidden layer
    (imputation.py:44-119) with injected draws against the oracle's sequential sweep -- ten prior draws through ten
    5000 x 5000 factors, speculative batches of four, the log-likelihoods of the THREE nodes upstairs summed
    (imputation.py:91-106), accept / shrink on the device.  Same accepted latents, the uniforms consumed one per proposal,
    the accepted log-likelihood.  (~2 s of host LAPACK per proposal for the oracle.)"""
    from oracle import dgp_oracle as O
    from dgp_amd.imputation import imputer, DrawStream
    n, d, q = 5000, 10, 3
    rng = np.random.default_rng(2026)
    X = rng.uniform(size=(n, d))
    F0 = np.stack([np.sin(2.0 * X[:, k] + 0.3 * k) + 0.5 * X[:, (k + 1) % d] ** 2 for k in range(d)], 1)
    F0 = (F0 - F0.mean(0)) / F0.std(0) + 0.05 * rng.standard_normal((n, d))
    Y = np.stack([np.sin(1.0 / ((0.7 * X[:, 0] + 0.3) * (0.7 * X[:, 1] + 0.3))) + (0.2 + 0.1 * j) * (X[:, 2 + j:] ** 2).sum(1) for j in range(q)], 1)
    Y = (Y - Y.mean(0)) / Y.std(0)
    layer0 = [_node(eng, 'sexp', [1.0 + 0.05 * k], X, F0[:, k], nugget=1e-6) for k in range(d)]
    upper = [_node(eng, 'sexp', [1.5 + 0.2 * j], F0, Y[:, j], glob=X, scale=0.8 + 0.1 * j, scale_est=True, nugget=1e-4) for j in range(q)]
    z = [rng.standard_normal(n) for _ in range(d)]
    u = [1e-12] + list(rng.random(40))      # a low threshold: the update ends within a few proposals (bounds the oracle's time)
    imp = imputer([layer0, upper], block=True, draws=DrawStream(z=[v.copy() for v in z], u=list(u)), engine=eng, batch=4)
    imp.batch_next = 4
    nu = np.stack([O.fmvn(float(nd.scale[0]) * O.k_matrix(X, nd.length, nd.nugget[0], nd.name), z[k]) for k, nd in enumerate(layer0)], 1)

    def up(fp):
        Xi = np.concatenate((fp, X), 1)
        return sum(O.log_likelihood(Xi, Y[:, [j]], nd.length, float(nd.scale[0]), nd.nugget[0], nd.name) for j, nd in enumerate(upper))

    f_ref, nprop, thetas, lls, log_y = O.ess_block_sweep(F0, nu, up, np.log(u[0]), u[1:])
    imp.sample(burnin=0)
    F1 = np.stack([nd.output[:, 0] for nd in layer0], 1)
    assert len(imp.draws._ubuf) == len(u) - (1 + nprop), 'uniforms consumed: threshold + one per proposal'
    assert imp.stats['proposals'] == nprop
    close(F1, f_ref, rtol=1e-8, atol=1e-10)
    close(imp._ll_cache[0], lls[-1], rtol=1e-9)


def test_cfg3_sexp_link_gp_at_full_size_vs_oracle(eng):
    """The SExp linked-GP predictor (functions.py:396-451: IJ_sexp's terms formed in flight on f64 MFMA) of a cfg3 second-
    layer node at n = 5000 -- ten uncertain local inputs, ten deterministic global ones -- against the oracle's direct
    evaluation for eight test points, from the SAME R^-1 and R^-1 y (uploaded), so that only the pair kernel is compared:
    means 1e-9, variances 1e-8 of the scale."""
    from oracle import dgp_oracle as O
    n, Dw, Dz, M = 5000, 10, 10, 1152   # nine 128-point workgroup chunks on the device; 64 of the points walked by the oracle
    rng = np.random.default_rng(5)
    W, Wg = rng.normal(size=(n, Dw)), rng.uniform(size=(n, Dz))
    y = np.sin(W[:, 0]) + Wg[:, 1] ** 2 + 0.1 * rng.normal(size=n)
    length, scale, nugget = np.array([2.5]), 1.3, 1e-4
    st = O.compute_stats(np.concatenate((W, Wg), 1), y[:, None], length, nugget, 'sexp', Dw)
    m, v = rng.normal(size=(M, Dw)), rng.uniform(0.01, 0.4, size=(M, Dw))
    v[0, 0] = 0.0
    z = rng.uniform(size=(M, Dz))
    # the first eight points by the reference's own pair formula, 56 more -- both sides of every 128-point chunk boundary and a
    # random rest -- by its GEMM form (oracle.IJ_sexp_gemm, pinned on the direct form in tests/test_oracle_golden.py)
    edge = [c for b in range(128, M, 128) for c in (b - 1, b)] + [M - 1]
    pick = np.array(sorted(set(range(8)) | set(edge) | set(rng.choice(M, 64, replace=False).tolist()))[:64])
    slow = pick[pick < 8]
    mo8, vo8 = O.link_gp_predict(m[slow], v[slow], z[slow], W, Wg, st['Rinv'], st['Rinv_y'], scale, length, nugget, 'sexp')
    from oracle_pool import link_gp_predict as oracle_link
    mo, vo = oracle_link(m[pick], v[pick], z[pick], W, Wg, st['Rinv'], st['Rinv_y'], scale, length, nugget, 'sexp', gemm_form=True, workers=8)
    close(mo[:len(slow)], mo8, rtol=1e-11, atol=1e-13)
    close(vo[:len(slow)], vo8, rtol=1e-10, atol=1e-11 * scale)
    lm, lv = eng.linkgp_predict('sexp', eng.tensor(m), eng.tensor(v), eng.tensor(z), eng.tensor(W), eng.tensor(Wg), length,
                                eng.tensor(st['Rinv']), n, eng.tensor(st['Rinv_y']), scale, nugget)
    lm, lv = npy(lm), npy(lv)
    assert len(pick) == 64 and np.all(np.isfinite(lm)) and np.all(np.isfinite(lv))
    close(lm[pick], mo, rtol=1e-9, atol=1e-11)
    close(lv[pick], vo, rtol=1e-8, atol=1e-8 * scale)


def test_cfg2_matern_link_gp_at_full_size_vs_oracle(eng):
    """The kernel that is four fifths of the headline's prediction time, at the headline's own shape: the Matern-2.5 linked-GP
    predictor (functions.py:396-430,453-494; vecchia.py:915-988 for the closed forms of Jd) of cfg2's output node -- n = 2000,
    five uncertain local inputs, five deterministic global ones -- with the training points in the order the emulator hands
    them over (Engine.linkgp_cells -> ops.cell_order, so the pair kernel's order classes are in play) and M = 2304 test
    points: two 1024-point launches and a ragged 256-point one.  Both sides take the DEVICE's R^-1 and R^-1 y (downloaded),
    so only the record / pair / finalize kernels are compared.  The oracle walks 32-36 points: both sides of every launch
    boundary and of 32-point workgroup boundaries next to them, the first and last points, a point with one zero input
    variance, one with all variances zero, one whose mean lies exactly on a training coordinate (in a v > 0 and in a v = 0
    dimension), random others.  Means 1e-9, variances 1e-8 of the scale.  (~1.5 s of oracle per point.)"""
    from oracle import dgp_oracle as O
    n, Dw, Dz, M = 2000, 5, 5, 2304
    rng = np.random.default_rng(2026)
    W = rng.normal(size=(n, Dw))                    # a hidden layer's latents: standard-normal-ish columns
    Wg = rng.uniform(size=(n, Dz))
    y = np.sin(W[:, 0]) + 0.5 * W[:, 1] * Wg[:, 0] + Wg[:, 2] ** 2 + 0.05 * rng.normal(size=n)
    y = (y - y.mean()) / y.std()
    length, scale, nugget = np.array([1.7]), 1.3, 1e-4      # one shared lengthscale, as the bench's nodes (functions.py:402-410)
    Xd = eng.tensor(np.concatenate((W, Wg), 1))
    A = eng.kmatrix('matern2.5', Xd, None, None, length, nugget, full=False, Y=eng.tensor(y))
    _, info = eng.potrf(n, A)
    work = eng.potrf_workspace(n, 1)
    Np = eng.padded_dim(n)
    Ainv = eng.empty(Np, Np)
    eng.potri(n, A, Ainv, 1, work)
    assert int(npy(info)[0]) == 0
    ry = (-Ainv[n, :n]).contiguous()
    cells = eng.linkgp_cells('matern2.5', W, eng.tensor(Wg), Ainv, ry)
    assert cells is not None
    Wc, Wgc = npy(cells['W']), npy(cells['Wg'])
    Ri, ryc = npy(cells['Rinv'])[:n, :n].copy(), npy(cells['ry'])
    assert sorted(map(tuple, Wc)) == sorted(map(tuple, W))            # a permutation of the training points
    m = rng.normal(size=(M, Dw))
    v = 10.0 ** rng.uniform(-4, -0.5, size=(M, Dw))
    z = rng.uniform(size=(M, Dz))
    v[5, 2] = 0.0                       # one deterministic dimension among uncertain ones (the v = 0 branch of IJ_matern)
    v[1029] = 0.0                       # all of them: J = outer product of the correlations
    m[1023, 1] = Wc[777, 1]             # a mean exactly on a training coordinate, v > 0 (the x_i = x_j = mu corner of Jd / Jd0)
    m[2050, 3], v[2050, 3] = Wc[13, 3], 0.0     # ... and in a v = 0 dimension
    edge = [0, 1, 5, 31, 32, 1023, 1024, 1029, 1055, 1056, 2047, 2048, 2050, 2079, 2080, 2271, 2272, M - 2, M - 1]
    pick = np.array(sorted(set(edge) | set(rng.choice(M, 17, replace=False).tolist())))
    from oracle_pool import link_gp_predict as oracle_link        # (the oracle's own function, its test points dealt to host processes)
    mo, vo = oracle_link(m[pick], v[pick], z[pick], Wc, Wgc, Ri, ryc, scale, length, nugget, 'matern2.5')
    lm, lv = eng.linkgp_predict('matern2.5', eng.tensor(m), eng.tensor(v), eng.tensor(z), cells['W'], cells['Wg'], length,
                                cells['Rinv'], Np, cells['ry'], scale, nugget)
    lm, lv = npy(lm), npy(lv)
    assert len(pick) >= 32 and np.all(np.isfinite(lm)) and np.all(np.isfinite(lv))
    close(lm[pick], mo, rtol=1e-9, atol=1e-9)
    close(lv[pick], vo, rtol=1e-8, atol=1e-8 * scale)
    # the caller's order (no cells: every 16-row group takes the two-product path) gives the same predictions
    um, uv = eng.linkgp_predict('matern2.5', eng.tensor(m), eng.tensor(v), eng.tensor(z), eng.tensor(W), eng.tensor(Wg), length,
                                Ainv, Np, ry, scale, nugget)
    close(npy(um)[pick], mo, rtol=1e-9, atol=1e-9)
    close(npy(uv)[pick], vo, rtol=1e-8, atol=1e-8 * scale)


def test_cfg2_emulator_predict_of_the_bench_model_vs_oracle_walk(eng):
    """emulator.predict (emulation.py:701-779,846-847) of the MODEL bench.py times -- configs[1]: n = 2000, d = 5, five
    Matern-2.5 nodes feeding one Matern-2.5 output node with the global connection, after two SI iterations -- on 1040 points
    (one full 1024-point launch of the pair kernel and a ragged one) and two imputations, against the oracle's walk of the
    same layers for 16 of the points: functions.gp at the inputs for every first-layer node, functions.link_gp through the
    output node per imputation, the mixture's moments at the end.  The model keeps the reference's default nugget 1e-6, so at
    n = 2000 R^-1 y has entries of 1e2-1e4 and every predictive mean is a sum of 2000 terms that cancel to O(1): what two
    correct evaluations can agree to is set by the size of those sums, not by the result (the cfg5 default-nugget test's
    argument).  Hence: the oracle is fed the device's own R^-1 / R^-1 y of every node and, layer by layer, the device's own
    moments of the layer below (fed its OWN first-layer moments the oracle's output mean moved by 2e-8: the propagation of
    1e-11 differences through R^-1 y, not a kernel's error).  Bounds stated in the magnitudes of the sums (|R^-1 y|_1 = 5e5,
    |R^-1|_1 = 7e9 here) would be vacuous -- 1e-7 for a mean, 1e-2 for a variance; what the kernels achieve is recorded by
    the test (gpurun_out/diag_cfg2_emulator_vs_oracle.txt: means within 3e-11, variances within 3e-8) and the thresholds
    sit a factor 30 above that: means 1e-9, variances 1e-6 of the scale."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import dgp_oracle as O
    from oracle_pool import link_gp_predict as oracle_link
    from dgp_amd import emulator
    import bench
    n, d, S, M = 2000, 5, 2, 1040
    model, X, Y = bench.build_model(n, d, 100, 0)
    model.train(N=2, ess_burn=10, disable=True)
    emu = emulator(model.estimate(burnin=0), N=S, seed=7, device=0)
    xt = np.random.default_rng(5).uniform(size=(M, d))
    mu, var = emu.predict(xt)
    mu_s, var_s = emu.predict(xt, aggregation=False)
    assert mu.shape == (M, 1) and np.all(np.isfinite(mu)) and np.all(np.isfinite(var))
    layers = [(npy(a), npy(b)) for a, b in emu._layer_moments(xt)]           # per layer (S, M, K) means and variances as the device walked them
    pick = np.array([0, 1, 2, 31, 32, 500, 777, 1000, 1022, 1023, 1024, 1025, 1030, 1037, 1038, 1039])
    xp = xt[pick]
    l1, out = emu.all_layer[0], emu.all_layer[1][0]
    diag = []
    mus, vs = [], []
    for s in range(S):
        for k, nd in enumerate(l1):
            st = emu._stats[(0, k)]
            Ri, ry = npy(st['Rinv'])[:n, :n], npy(st['ry'][s])
            mo, vo = O.gp_predict(xp, npy(st['Wall']), Ri, ry, nd.scale, nd.length, nd.nugget, nd.name)
            bm, bv = 1e-9, 1e-6 * float(nd.scale[0])
            diag.append(('gp s%d k%d' % (s, k), np.abs(layers[0][0][s][pick, k] - mo).max(), bm, np.abs(layers[0][1][s][pick, k] - vo).max(), bv))
            close(layers[0][0][s][pick, k], mo, rtol=1e-9, atol=bm)
            close(layers[0][1][s][pick, k], vo, rtol=1e-9, atol=bv)
        st = emu._stats[(1, 0)]
        ps = st['per'][s]
        Wg = ps.get('Wg', st['Wg'])
        Ri, ry = npy(ps['Rinv'])[:n, :n], npy(ps['ry'])
        pm, pv = layers[0][0][s][pick][:, out.input_dim], layers[0][1][s][pick][:, out.input_dim]      # the device's own first-layer moments
        mo, vo = oracle_link(pm, pv, xp[:, out.connect], npy(ps['W']), npy(Wg), Ri, ry, out.scale, out.length, out.nugget, out.name)
        bm, bv = 1e-9, 1e-6 * float(out.scale[0])
        diag.append(('link s%d' % s, np.abs(np.asarray(mu_s[s])[pick, 0] - mo).max(), bm, np.abs(np.asarray(var_s[s])[pick, 0] - vo).max(), bv))
        close(np.asarray(mu_s[s])[pick, 0], mo, rtol=1e-9, atol=bm)
        close(np.asarray(var_s[s])[pick, 0], vo, rtol=1e-9, atol=bv)
        assert np.array_equal(np.asarray(mu_s[s])[:, 0], layers[1][0][s][:, 0])      # (predict(aggregation=False) is that walk)
        mus.append(np.asarray(mu_s[s])[pick, 0])
        vs.append(np.asarray(var_s[s])[pick, 0])
    # the mixture over the imputations (emulation.py:846-847), from the device's own per-imputation moments: exact arithmetic of a few terms
    mo, vo = O.aggregate_moments(mus, vs)
    close(mu[pick, 0], mo, rtol=1e-12, atol=1e-14)
    close(var[pick, 0], vo, rtol=1e-10, atol=1e-13)
    root = os.environ.get('GRAFT_REPO_ROOT')
    if root and os.path.isdir(os.path.join(root, 'gpurun_out')):      # (how far inside its bounds every comparison was: kept with the run's records)
        with open(os.path.join(root, 'gpurun_out', 'diag_cfg2_emulator_vs_oracle.txt'), 'w') as f:
            f.write('what | max |mean diff| | bound | max |var diff| | bound\n')
            for row in diag:
                f.write('%-10s %.3e %.3e %.3e %.3e\n' % row)


def test_cfg5_chain_at_full_size_vs_oracle_walk(eng):
    """BASELINE configs[4] at its own size: GP -> DGP -> GP, n = 1000 each, Matern-2.5, through the public lgp.predict
    (linkgp.py:285-501) against the oracle's walk of the same chain -- gp at the inputs, then per imputation link_gp
    through the DGP's two layers and through the last GP, the mixture's moments over the imputations at the end
    (linkgp.py:495-500) -- 256 test points and two imputations on the device, the first 16 points walked by the oracle (its
    Matern link_gp takes ~0.4 s per point and node).  (The DGP's layers are chained without a global connection: the
    oracle has no linkgp_prediction_full; that branch is pinned on the reference's own output at small n, g10.)"""
    from oracle import dgp_oracle as O
    from dgp_amd.linkgp import container, lgp
    n, M, S = 1000, 256, 2
    rng = np.random.default_rng(9)
    X1 = rng.uniform(size=(n, 3))
    Y1 = np.sin(3 * X1[:, 0]) + X1[:, 1] ** 2 - X1[:, 2]
    Y1 = (Y1 - Y1.mean()) / Y1.std()
    Y2 = np.tanh(2 * Y1) + 0.3 * Y1 ** 2
    Y2 = (Y2 - Y2.mean()) / Y2.std()
    Y3 = np.cos(2 * Y2)
    Y3 = (Y3 - Y3.mean()) / Y3.std()
    # (nuggets that keep cond(R) ~ 1e5 with 1000 points on a line: at 1e-6 both sides lose five digits of R^-1 y to
    #  conditioning, each in its own way, and the comparison says nothing)
    g1 = _node(eng, 'matern2.5', [0.8, 1.2, 1.0], X1, Y1, scale=1.1, nugget=1e-3)
    g3 = _node(eng, 'matern2.5', [1.0], Y2[:, None], Y3, scale=0.9, nugget=1e-2)
    sets = []
    lat = [np.tanh(1.5 * Y1) + 0.05 * rng.standard_normal(n) for _ in range(S)]     # the imputations' hidden layers
    for s in range(S):
        h = _node(eng, 'matern2.5', [1.1], Y1[:, None], lat[s], scale=1.0, nugget=1e-2)
        t = _node(eng, 'matern2.5', [0.9], lat[s][:, None], Y2, scale=1.2, nugget=1e-2)
        one = []
        for l, st in enumerate(([[g1]], [[h], [t]], [[g3]])):
            c = container.__new__(container)
            c.vecch, c.local_input_idx = False, (np.array([0, 1, 2]) if l == 0 else np.array([0]))
            if len(st) == 1:
                c.type, c.structure = 'gp', st[0][0]
            else:
                c.type, c.structure = 'dgp', st
            one.append([c])
        sets.append(one)
    sysm = lgp.__new__(lgp)
    sysm.L, sysm.all_layer, sysm.num_model, sysm.all_layer_set = 3, sets[0], [1, 1], sets
    xt = rng.uniform(size=(M, 3))
    mu, var = sysm.predict([xt, [None], [None]])

    def stats(nd):
        Xn = nd.input if nd.global_input is None else np.concatenate((nd.input, nd.global_input), 1)
        return O.compute_stats(Xn, nd.output, nd.length, nd.nugget[0], nd.name, nd.input.shape[1])

    from oracle_pool import link_gp_predict as oracle_link

    def link(nd, m, v):
        st = stats(nd)
        return oracle_link(m[:, None], v[:, None], None, nd.input, None, st['Rinv'], st['Rinv_y'], nd.scale, nd.length, nd.nugget, nd.name)

    # 24 of the 256 points (the ends, both sides of the 128-point chunk boundary, four more): ~2.5 s of oracle time per point
    pick = np.array(sorted(set(range(4)) | set(range(122, 134)) | set(range(252, 256)) | {37, 77, 181, 219}))
    K = len(pick)
    s1 = stats(g1)
    m1, v1 = O.gp_predict(xt[pick], X1, s1['Rinv'], s1['Rinv_y'], g1.scale, g1.length, g1.nugget, g1.name)
    mus, vars_ = [], []
    for s in range(S):
        h, t = sets[s][1][0].structure[0][0], sets[s][1][0].structure[1][0]
        mh, vh = link(h, m1, v1)
        mt, vt = link(t, mh, vh)
        m3s, v3s = link(g3, mt, vt)
        mus.append(m3s)
        vars_.append(v3s)
    m3 = np.mean(mus, 0)
    v3 = np.mean([a ** 2 + b for a, b in zip(mus, vars_)], 0) - m3 ** 2
    assert mu[0].shape == (M, 1) and np.all(np.isfinite(mu[0])) and np.all(var[0] > -1e-10)
    assert K == 24
    close(mu[0][pick, 0], m3, rtol=1e-6, atol=1e-8)
    close(var[0][pick, 0], v3, rtol=1e-5, atol=1e-7)


def test_cfg5_chain_default_nugget_backward_error_form(eng):
    """BASELINE configs[4] at its own size AND at the reference's default nugget 1e-6 (kernel_class.py:34: cond(R) ~ 1e8 with
    1000 points on a line).  A forward comparison of predictions says nothing there -- LAPACK and the device each lose eight
    digits of R^-1 y in their own way (the test above raises the nuggets for that reason) -- so the chain GP -> DGP -> GP is
    checked in backward-error form: (1) every node's device statistics solve their systems to working accuracy,
    |R (R^-1 y) - y| and |R R^-1 - I| small relative to |R| |R^-1| (what a backward-stable solver guarantees, whatever the
    conditioning); (2) node by node -- gp for the first GP, link_gp for the DGP's two layers and the last GP
    (linkgp.py:285-501 walks exactly these) -- the device's predictor against the oracle's fed with THE DEVICE'S R^-1 and
    R^-1 y and the same inputs, within bounds stated in the magnitudes of the sums involved; lgp.predict of the whole chain
    runs and stays finite."""
    from oracle import dgp_oracle as O
    from dgp_amd.linkgp import container, lgp
    n, M = 1000, 64
    rng = np.random.default_rng(19)
    X1 = rng.uniform(size=(n, 3))
    Y1 = np.sin(3 * X1[:, 0]) + X1[:, 1] ** 2 - X1[:, 2]
    Y1 = (Y1 - Y1.mean()) / Y1.std()
    Y2 = np.tanh(2 * Y1) + 0.3 * Y1 ** 2
    Y2 = (Y2 - Y2.mean()) / Y2.std()
    Y3 = np.cos(2 * Y2)
    Y3 = (Y3 - Y3.mean()) / Y3.std()
    lat = np.tanh(1.5 * Y1) + 0.05 * rng.standard_normal(n)
    g1 = _node(eng, 'matern2.5', [0.8, 1.2, 1.0], X1, Y1, scale=1.1)          # nugget 1e-6: the default
    h = _node(eng, 'matern2.5', [1.1], Y1[:, None], lat, scale=1.0)
    t = _node(eng, 'matern2.5', [0.9], lat[:, None], Y2, scale=1.2)
    g3 = _node(eng, 'matern2.5', [1.0], Y2[:, None], Y3, scale=0.9)
    one = []
    for l, st in enumerate(([[g1]], [[h], [t]], [[g3]])):
        c = container.__new__(container)
        c.vecch, c.local_input_idx = False, (np.array([0, 1, 2]) if l == 0 else np.array([0]))
        if len(st) == 1:
            c.type, c.structure = 'gp', st[0][0]
        else:
            c.type, c.structure = 'dgp', st
        one.append([c])
    sysm = lgp.__new__(lgp)
    sysm.L, sysm.all_layer, sysm.num_model, sysm.all_layer_set = 3, one, [1, 1], [one]
    xt = rng.uniform(size=(M, 3))
    mu, var = sysm.predict([xt, [None], [None]])
    assert mu[0].shape == (M, 1) and np.all(np.isfinite(mu[0])) and np.all(np.isfinite(var[0]))
    dev = {}
    for nd in (g1, h, t, g3):
        if nd._stats is None:
            nd.compute_stats()
        R = O.k_matrix(nd.input, nd.length, nd.nugget[0], nd.name)
        Ri, ry = nd.Rinv, nd.Rinv_y
        y = nd.output[:, 0]
        nR, nRi = np.abs(R).sum(1).max(), np.abs(Ri).sum(1).max()
        assert nR * nRi > 1e6          # (the conditioning this test is about)
        assert np.abs(R @ ry - y).max() <= 1e-11 * (nR * np.abs(ry).max() + np.abs(y).max())
        assert np.abs(R @ Ri - np.eye(n)).max() <= 1e-11 * nR * nRi
        assert np.abs(Ri - Ri.T).max() <= 1e-12 * nRi
        dev[id(nd)] = (Ri, ry)
    # (2) node by node, the same inputs and the same R^-1 / R^-1 y on both sides.  What is left is the order of sums whose terms
    # are as large as |R^-1 y|_1 (mean) and |R^-1 y|_1^2 + |R^-1|_1 (variance: y'R^-1 J R^-1 y - tr(R^-1 J) with |J| <= 1), so
    # the bounds are stated in those magnitudes -- at this conditioning they are 1e5 .. 1e12, which is also why the three-node
    # chain as a whole cannot be compared forward (each implementation's 1e-13 relative differences in I and J come out of
    # these sums as 1e-3 of a variance and grow from node to node; the reference's own LAPACK run has the same property).
    K = 12   # (the oracle's Matern link_gp takes ~0.4 s per point and node)
    Ri, ry = dev[id(g1)]
    mo, vo = O.gp_predict(xt[:K], X1, Ri, ry, g1.scale, g1.length, g1.nugget, g1.name)
    md, vd = g1.gp_prediction(xt[:K], None)
    close(np.ravel(md), np.ravel(mo), rtol=1e-9, atol=1e-12 * np.abs(ry).sum())
    close(np.ravel(vd), np.ravel(vo), rtol=1e-9, atol=1e-12 * float(g1.scale[0]) * np.abs(Ri).sum())
    for nd, lo, hi in ((h, Y1.min(), Y1.max()), (t, lat.min(), lat.max()), (g3, Y2.min(), Y2.max())):
        Ri, ry = dev[id(nd)]
        m_in = rng.uniform(lo, hi, size=(K, 1))
        v_in = rng.uniform(1e-4, 5e-2, size=(K, 1))
        mo, vo = O.link_gp_predict(m_in, v_in, None, nd.input, None, Ri, ry, nd.scale, nd.length, nd.nugget, nd.name)
        md, vd = nd.linkgp_prediction(m_in, v_in, None)
        s1 = np.abs(ry).sum()
        close(np.ravel(md), np.ravel(mo), rtol=1e-9, atol=1e-12 * s1)
        close(np.ravel(vd), np.ravel(vo), rtol=1e-9, atol=1e-12 * float(nd.scale[0]) * (s1 * s1 + np.abs(Ri).sum()))


def test_cfg4_vecchia_prediction_at_full_size_vs_oracle(eng):
    """cfg4's prediction kernels at their own size (n = 50 000 training points, 50 neighbours, 20 000 test points): gp_vecch
    (vecchia.py:635-654, D = 8) and link_gp_vecch (:758-796, Dw = Dz = 8, squared exponential) through the register-resident
    kernels, twelve test points spread over the launch against the oracle's per-point Cholesky, the neighbour rows of those
    points against brute force; every output finite and the variances positive."""
    from oracle import dgp_oracle as O
    rng = np.random.default_rng(404)
    n, M, pm = 50000, 20000, 50
    X = rng.uniform(size=(n, 8))
    W = np.sin(3 * X) + 0.1 * rng.normal(size=(n, 8))
    y = np.sin(X.sum(1)) + 0.05 * rng.normal(size=n)
    xq = rng.uniform(size=(M, 8))
    mm = np.sin(3 * xq) + 0.02 * rng.normal(size=(M, 8))
    vv = rng.uniform(0.001, 0.05, size=(M, 8))
    ones = np.ones(n)
    l1, l2 = np.array([0.6]), np.array([1.5])
    NN1 = eng.nn_query(eng.tensor(xq / l1), eng.tensor(X / l1), pm)
    A2, Q2 = np.concatenate((W, X), 1), np.concatenate((mm, xq), 1)
    NN2 = eng.nn_query(eng.tensor(Q2 / l2), eng.tensor(A2 / l2), pm)
    gm, gv = (npy(t) for t in eng.vecchia_gp('sexp', eng.tensor(xq), eng.tensor(X), NN1, eng.tensor(y), 1.2, l1, 1e-4, eng.tensor(ones)))
    lm, lv = (npy(t) for t in eng.vecchia_linkgp('sexp', eng.tensor(mm), eng.tensor(vv), eng.tensor(xq), eng.tensor(W), eng.tensor(X), NN2,
                                                 eng.tensor(y), 1.2, l2, 1e-4, eng.tensor(ones)))
    assert np.all(np.isfinite(gm)) and np.all(gv > 0) and np.all(np.isfinite(lm)) and np.all(lv >= 0)
    pick = np.array([0, 1, 2, 3, 777, 5000, 9999, 10001, 15000, 19997, 19998, 19999])
    N1, N2 = npy(NN1).astype(int)[pick], npy(NN2).astype(int)[pick]
    for i, t in enumerate(pick):   # the neighbour rows: brute force, nearest first
        d = ((X - xq[t]) ** 2).sum(1)
        assert set(N1[i].tolist()) == set(np.argsort(d, kind='stable')[:pm].tolist())
        d = ((A2 - Q2[t]) ** 2).sum(1)
        assert set(N2[i].tolist()) == set(np.argsort(d, kind='stable')[:pm].tolist())
    mo, vo = O.gp_vecch(xq[pick], X, N1, y, 1.2, l1, 1e-4, ones, 'sexp')
    close(gm[pick], mo, rtol=1e-8, atol=1e-10)
    close(gv[pick], vo, rtol=1e-6, atol=1e-10)
    mo, vo = O.link_gp_vecch(mm[pick], vv[pick], xq[pick], W, X, N2, y, 1.2, l2, 1e-4, ones, 'sexp')
    close(lm[pick], mo, rtol=1e-8, atol=1e-10)
    close(lv[pick], vo, rtol=1e-6, atol=1e-9)
