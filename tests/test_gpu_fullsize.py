"""Size-independent properties of the HIP path at BASELINE.json's full sizes (the oracle is too slow there):
cfg3's n = 5000 factorisation and inverse, cfg2's n = 2000 predictors, cfg4's n = 50 000 Vecchia kernels.
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
