"""CPU oracle for the dgpsi stochastic-imputation hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this module; dgp_amd/ never does.

A vectorised numpy/scipy restatement of the algorithm of mingdeyu/DGP (dgpsi
2.6.0).  Each function cites the reference file:line it follows.  It is pinned
against golden vectors recorded from the reference itself (run in the build
container under identity stubs of numba/pathos, see oracle/gen_golden.py and
tests/golden/*.npz; tests/test_oracle_golden.py).

Conventions: every array is float64 C-order; index arrays are int64; a GP node
sees X = [input | global_input] (kernel_class.py:318-322); `length` has either
one entry (shared) or one per column of X; kernel names are the reference's
strings 'sexp' / 'matern2.5'.
"""
import numpy as np
from scipy.linalg import cholesky, cho_factor, cho_solve, solve_triangular
from scipy.special import erf

SQ5 = np.sqrt(5.0)


# --------------------------------------------------------------------------
# a1/a2  kernel matrices (kernel_class.py:304-359, functions.py:16-93)
# --------------------------------------------------------------------------
def _absdiff(Xl, d):
    c = Xl[:, d]
    return np.abs(c[:, None] - c[None, :])


def corr_matrix(X, length, name):
    """Correlation matrix WITHOUT the nugget diagonal (unit diagonal).

    sexp:      exp(-sum_d ((x_id-x_jd)/g_d)^2)            kernel_class.py:324-327
    matern2.5: prod_d(1+sqrt5 r+5/3 r^2) exp(-sqrt5 sum r) kernel_class.py:343-345,
               functions.py:28-34 (separable product form)
    """
    X = np.asarray(X, dtype=float)
    Xl = X / np.asarray(length, dtype=float)
    n, D = Xl.shape
    if name == 'sexp':
        s = np.zeros((n, n))
        for d in range(D):
            df = Xl[:, d][:, None] - Xl[:, d][None, :]
            s += df * df
        return np.exp(-s)
    elif name == 'matern2.5':
        prod = np.ones((n, n))
        s = np.zeros((n, n))
        for d in range(D):
            r = _absdiff(Xl, d)
            prod *= 1.0 + SQ5 * r + (5.0 / 3.0) * r * r
            s += r
        return prod * np.exp(-SQ5 * s)
    raise ValueError(name)


def k_matrix(X, length, nugget, name, W_diag=None):
    """kernel.k_matrix(fod_eval=False): diagonal overwritten with 1+nugget (or
    1+nugget*W_diag with replicates).  kernel_class.py:352-355."""
    K = corr_matrix(X, length, name)
    n = K.shape[0]
    w = np.ones(n) if W_diag is None else np.asarray(W_diag, dtype=float)
    K[np.arange(n), np.arange(n)] = 1.0 + float(nugget) * w
    return K


def k_matrix_fod(X, length, nugget, name, nugget_est, W_diag=None):
    """kernel.k_matrix(fod_eval=True) -> (K, fod[p,n,n]); p = len(length)(+1).

    dK/dlog g: sexp 2 r_d^2 K (shared: 2 sum_d r_d^2 K)  kernel_class.py:328-332,
    functions.py:36-45; matern c_d K, c_d=(5/3) r^2 (1+sqrt5 r)/(1+sqrt5 r+5/3 r^2)
    functions.py:71-93.  dK/dlog eta = eta*I (eta*diag(W)) kernel_class.py:346-351.
    """
    X = np.asarray(X, dtype=float)
    length = np.asarray(length, dtype=float)
    Xl = X / length
    n, D = Xl.shape
    K = corr_matrix(X, length, name)
    shared = (len(length) == 1)
    P = 1 if shared else D
    fod = np.zeros((P, n, n))
    for d in range(D):
        r = _absdiff(Xl, d)
        if name == 'sexp':
            c = 2.0 * r * r
        else:
            e1 = 1.0 + SQ5 * r
            e2 = (5.0 / 3.0) * r * r
            c = e2 * e1 / (e1 + e2)
        fod[0 if shared else d] += c * K
    w = np.ones(n) if W_diag is None else np.asarray(W_diag, dtype=float)
    if nugget_est:
        fod = np.concatenate((fod, (float(nugget) * np.diag(w))[None]), axis=0)
    K[np.arange(n), np.arange(n)] = 1.0 + float(nugget) * w
    return K, fod


# --------------------------------------------------------------------------
# priors (kernel_class.py:93-110,361-401, functions.py:95-100)
# --------------------------------------------------------------------------
def log_prior(length, nugget, prior_name, prior_coef, nugget_est, cl=None):
    """prior_coef is the STORED coefficient (ga: shape-1; inv_ga: shape+1)."""
    length = np.asarray(length, dtype=float)
    nugget = float(np.ravel(nugget)[0])
    if prior_name is None:
        return 0.0
    if prior_name == 'ref':
        a, b = prior_coef[0], prior_coef[1]
        t = np.sum(cl / length) + nugget
        return a * np.log(t) - b * t
    a, b = prior_coef[0], prior_coef[1]
    xs = length if not nugget_est else np.concatenate((length, [nugget]))
    if prior_name == 'ga':
        return float(np.sum(a * np.log(xs) - b * xs))
    return float(np.sum(-a * np.log(xs) - b / xs))


def log_prior_fod(length, nugget, prior_name, prior_coef, nugget_est, cl=None):
    length = np.asarray(length, dtype=float)
    nugget = float(np.ravel(nugget)[0])
    if prior_name == 'ref':
        a, b = prior_coef[0], prior_coef[1]
        t = np.sum(cl / length) + nugget
        fod = (b - a / t) * cl / length
        if nugget_est:
            fod = np.concatenate((np.atleast_1d(fod), [(a / t - b) * nugget]))
        return np.atleast_1d(fod)
    a, b = prior_coef[0], prior_coef[1]
    xs = length if not nugget_est else np.concatenate((length, [nugget]))
    if prior_name == 'ga':
        return a - b * xs
    return -a + b / xs


def compute_cl(X, length_len, n_out, vecch=False):
    """kernel.compute_cl, kernel_class.py:207-225 (X = [input|global_input])."""
    X = np.asarray(X, dtype=float)
    if length_len == 1:
        if vecch:
            rg = X.max(0) - X.min(0)
            return np.sqrt(rg @ rg) / n_out
        from scipy.spatial.distance import pdist
        return np.max(pdist(X, metric='euclidean')) / n_out
    rg = X.max(0) - X.min(0)
    return rg / n_out ** (1.0 / length_len)


# --------------------------------------------------------------------------
# a5  ESS target log-likelihood (kernel_class.py:481-492)
# --------------------------------------------------------------------------
def log_likelihood(X, y, length, scale, nugget, name, W_diag=None):
    """-0.5 (logdet(s2 K) + y^T (s2 K)^-1 y); no 2pi term, no prior."""
    cov = float(np.ravel(scale)[0]) * k_matrix(X, length, nugget, name, W_diag)
    L = cholesky(cov, lower=True, check_finite=False)
    logdet = 2.0 * np.sum(np.log(np.abs(np.diag(L))))
    y = np.asarray(y, dtype=float).reshape(-1, 1)
    quad = float((y.T @ cho_solve((L, True), y, check_finite=False))[0, 0])
    return -0.5 * (logdet + quad)


# --------------------------------------------------------------------------
# a8  M-step objective and gradient (kernel_class.py:403-449)
# --------------------------------------------------------------------------
def nll_grad(x, X, y, name, scale, nugget, nugget_est, scale_est,
             prior_name='ga', prior_coef=None, cl=None,
             W_diag=None, n_rep=None, sum_residual=None):
    """Reference formulation (cho_solve with each dK as right-hand side).

    x = log(length) [+ log(nugget) iff nugget_est].  Returns (nll, grad, scale)
    where `scale` is the closed-form update when scale_est (side effect of
    kernel.llik, kernel_class.py:428-431).  n_rep = len(rep) with replicates.
    """
    theta = np.exp(np.asarray(x, dtype=float))
    if nugget_est:
        length, nugget = theta[:-1], theta[-1]
    else:
        length = theta
    nugget = float(np.ravel(nugget)[0])
    scale = float(np.ravel(scale)[0])
    y = np.asarray(y, dtype=float).reshape(-1, 1)
    n = y.shape[0]
    K, Kt = k_matrix_fod(X, length, nugget, name, nugget_est, W_diag)
    L = cholesky(K, lower=True, check_finite=False)
    KinvKt = np.array([cho_solve((L, True), Kt_i, check_finite=False) for Kt_i in Kt])
    tr = np.trace(KinvKt, axis1=1, axis2=2)
    logdet = 2.0 * np.sum(np.log(np.abs(np.diag(L))))
    KinvY = cho_solve((L, True), y, check_finite=False)
    YKKY = (y.T @ KinvKt @ KinvY).flatten()
    YKinvY = float((y.T @ KinvY)[0, 0])
    P1 = -0.5 * tr
    P2 = 0.5 * YKKY
    rep = n_rep is not None
    if scale_est:
        if not rep:
            scale = YKinvY / n
            nll = 0.5 * (logdet + n * np.log(scale))
        else:
            scale = (YKinvY + sum_residual / nugget) / n_rep
            nll = 0.5 * (logdet + n_rep * np.log(scale))
        g = -P1 - P2 / scale
        if rep and nugget_est:
            nll += 0.5 * (n_rep - n) * np.log(nugget)
            g[-1] += 0.5 * (-sum_residual / (scale * nugget) + (n_rep - n))
    else:
        nll = 0.5 * (logdet + YKinvY / scale)
        g = -P1 - P2 / scale
        if rep and nugget_est:
            nll += 0.5 * (sum_residual / (scale * nugget) + (n_rep - n) * np.log(nugget))
            g[-1] += 0.5 * (-sum_residual / (scale * nugget) + (n_rep - n))
    if prior_name is not None:
        nll = nll - log_prior(length, nugget, prior_name, prior_coef, nugget_est, cl)
        g = g - log_prior_fod(length, nugget, prior_name, prior_coef, nugget_est, cl)
    return float(nll), np.asarray(g, dtype=float), scale


# --------------------------------------------------------------------------
# a3/a4  sampler pieces (functions.py:103-121,203-208)
# --------------------------------------------------------------------------
def fmvn(cov, z):
    """L z with L = chol(cov); z is the injected N(0,I) draw (functions.py:113-121)."""
    L = np.linalg.cholesky(cov)
    return (L @ np.asarray(z, dtype=float).reshape(-1, 1)).flatten()


def update_f(f, nu, theta):
    return f * np.cos(theta) + nu * np.sin(theta)


def ess_angles(u_theta):
    """The proposal-angle sequence of imputation.py:81-82,115-119 assuming every
    proposal so far was rejected.  u_theta[0] draws theta0 = 2pi u; later entries
    are the U(0,1) draws that place theta inside the shrinking bracket
    (numpy uniform(lo,hi) = lo + (hi-lo) u).  It is a function of the uniforms
    alone, so a batch of B speculative proposals + 'first accepted' reproduces
    the sequential loop exactly."""
    u_theta = np.asarray(u_theta, dtype=float)
    out = np.empty_like(u_theta)
    theta = 2.0 * np.pi * u_theta[0]
    lo, hi = theta - 2.0 * np.pi, theta
    out[0] = theta
    for i in range(1, len(u_theta)):
        if theta < 0.0:
            lo = theta
        else:
            hi = theta
        theta = lo + (hi - lo) * u_theta[i]
        out[i] = theta
    return out


def ess_block_sweep(f, nu, upper_loglik, log_u0, u_theta):
    """One layer-wise ESS update (imputer.one_sample_block, imputation.py:44-119)
    with injected randomness.

    f, nu        : (n, M) current latent block and the prior draw (ellipse)
    upper_loglik : callable(fp (n,M)) -> sum of upper-layer log-likelihoods
    log_u0       : log of the threshold uniform (imputation.py:79)
    u_theta      : U(0,1) draws for the angle sequence (enough of them)
    Returns (f_new, n_proposals, thetas_tried, logliks_tried, log_y).
    """
    log_y = upper_loglik(f) + log_u0
    thetas = ess_angles(u_theta)
    lls = []
    for i, th in enumerate(thetas):
        fp = update_f(f, nu, th)
        ll = upper_loglik(fp)
        lls.append(ll)
        if ll > log_y:
            return fp, i + 1, thetas[:i + 1], np.array(lls), log_y
    raise RuntimeError('ess_block_sweep: ran out of injected uniforms')


# --------------------------------------------------------------------------
# a10  prediction statistics (kernel_class.py:735-764, functions.py:259-272)
# --------------------------------------------------------------------------
def compute_stats(X, y, length, nugget, name, n_local, W_diag=None):
    """-> dict(Rinv, Rinv_y, R2sexp, Psexp).  n_local = number of local (input)
    columns; R2sexp/Psexp are built on the local columns only (sexp only)."""
    R = k_matrix(X, length, nugget, name, W_diag)
    L = np.linalg.cholesky(R)
    n = len(R)
    Rinv = cho_solve((L, True), np.eye(n), check_finite=False)
    Rinv_y = cho_solve((L, True), np.asarray(y, float).reshape(-1, 1), check_finite=False).flatten()
    out = dict(Rinv=Rinv, Rinv_y=Rinv_y, R2sexp=None, Psexp=None)
    if name == 'sexp':
        length = np.asarray(length, dtype=float)
        ll = length if len(length) == 1 else length[:n_local]
        Xl = np.asarray(X, float)[:, :n_local] / ll
        s = np.zeros((n, n))
        for d in range(n_local):
            df = Xl[:, d][:, None] - Xl[:, d][None, :]
            s += df * df
        R2 = np.exp(-s / 2.0)
        R2[np.arange(n), np.arange(n)] = 1.0
        out['R2sexp'] = R2
        out['Psexp'] = np.stack([Xl[:, d][:, None] + Xl[:, d][None, :] for d in range(n_local)])
    return out


# --------------------------------------------------------------------------
# a11  GP prediction (functions.py:379-394, vecchia.py:244-265)
# --------------------------------------------------------------------------
def cross_corr(W, x, length, name):
    """r[i, t] = k(W_i, x_t): (n, M).  K_vec_nb, vecchia.py:244-265."""
    Wl = np.asarray(W, float) / np.asarray(length, float)
    xl = np.asarray(x, float) / np.asarray(length, float)
    n, D = Wl.shape
    M = xl.shape[0]
    if name == 'sexp':
        s = np.zeros((n, M))
        for d in range(D):
            df = Wl[:, d][:, None] - xl[:, d][None, :]
            s += df * df
        return np.exp(-s)
    prod = np.ones((n, M))
    s = np.zeros((n, M))
    for d in range(D):
        r = np.abs(Wl[:, d][:, None] - xl[:, d][None, :])
        prod *= 1.0 + SQ5 * r + (5.0 / 3.0) * r * r
        s += r
    return prod * np.exp(-SQ5 * s)


def gp_predict(x, W, Rinv, Rinv_y, scale, length, nugget, name):
    """m = Rinv_y . r ; v = |scale (1 + nugget - r^T Rinv r)|  (x, W already
    concatenated with the global columns).  functions.py:379-394."""
    r = cross_corr(W, x, length, name)
    Rr = Rinv @ r
    quad = np.sum(r * Rr, axis=0)
    m = Rinv_y @ r
    v = np.abs(float(np.ravel(scale)[0]) * (1.0 + float(np.ravel(nugget)[0]) - quad))
    return m, v


# --------------------------------------------------------------------------
# a12-a14  linked-GP prediction (functions.py:396-506, vecchia.py:838-1000)
# --------------------------------------------------------------------------
def _matern_point(d, ell):
    a = np.abs(d)
    return (1.0 + SQ5 * a / ell + 5.0 * d * d / (3.0 * ell * ell)) * np.exp(-SQ5 * a / ell)


def matern_I_dim(xk, zm, zv, ell):
    """E[k(x, Z)], Z~N(zm, zv), one dimension (functions.py:463-471)."""
    zX = zm - xk
    if zv == 0:
        return _matern_point(zX, ell)
    muA = zX - SQ5 * zv / ell
    muB = zX + SQ5 * zv / ell
    t1 = np.exp((5 * zv - 2 * SQ5 * ell * zX) / (2 * ell ** 2)) * (
        (1 + SQ5 * muA / ell + 5 * (muA ** 2 + zv) / (3 * ell ** 2)) * 0.5 * (1 + erf(muA / np.sqrt(2 * zv)))
        + (SQ5 + (5 * muA) / (3 * ell)) * np.sqrt(0.5 * zv / np.pi) / ell * np.exp(-0.5 * muA ** 2 / zv))
    t2 = np.exp((5 * zv + 2 * SQ5 * ell * zX) / (2 * ell ** 2)) * (
        (1 - SQ5 * muB / ell + 5 * (muB ** 2 + zv) / (3 * ell ** 2)) * 0.5 * (1 + erf(-muB / np.sqrt(2 * zv)))
        + (SQ5 - (5 * muB) / (3 * ell)) * np.sqrt(0.5 * zv / np.pi) / ell * np.exp(-0.5 * muB ** 2 / zv))
    return t1 + t2


def Jd(X1, X2, z_m, z_v, ell):
    """Off-diagonal Matern-2.5 J factor, vectorised over X1, X2 (vecchia.py:915-959)."""
    X1 = np.asarray(X1, float)
    X2 = np.asarray(X2, float)
    x1 = np.minimum(X1, X2)
    x2 = np.maximum(X1, X2)
    l = ell
    l4 = 9 * l ** 4
    pi = np.pi
    E30 = 1 + (25 * x1**2 * x2**2 - 3 * SQ5 * (3 * l**3 + 5 * l * x1 * x2) * (x1 + x2) + 15 * l**2 * (x1**2 + x2**2 + 3 * x1 * x2)) / l4
    E31 = (18 * SQ5 * l**3 + 15 * SQ5 * l * (x1**2 + x2**2) - (75 * l**2 + 50 * x1 * x2) * (x1 + x2) + 60 * SQ5 * l * x1 * x2) / l4
    E32 = 5 * (5 * x1**2 + 5 * x2**2 + 15 * l**2 - 9 * SQ5 * l * (x1 + x2) + 20 * x1 * x2) / l4
    E33 = 10 * (3 * SQ5 * l - 5 * x1 - 5 * x2) / l4
    E34 = 25 / l4
    muC = z_m - 2 * SQ5 * z_v / l
    E3A31 = E30 + muC * E31 + (muC**2 + z_v) * E32 + (muC**3 + 3 * z_v * muC) * E33 + (muC**4 + 6 * z_v * muC**2 + 3 * z_v**2) * E34
    E3A32 = E31 + (muC + x2) * E32 + (muC**2 + 2 * z_v + x2**2 + muC * x2) * E33 + (muC**3 + x2**3 + x2 * muC**2 + muC * x2**2 + 3 * z_v * x2 + 5 * z_v * muC) * E34
    P1 = np.exp((10 * z_v + SQ5 * l * (x1 + x2 - 2 * z_m)) / l**2) * (
        0.5 * E3A31 * (1 + erf((muC - x2) / np.sqrt(2 * z_v)))
        + E3A32 * np.sqrt(0.5 * z_v / pi) * np.exp(-0.5 * (x2 - muC)**2 / z_v))

    E40 = 1 + (25 * x1**2 * x2**2 + 3 * SQ5 * (3 * l**3 - 5 * l * x1 * x2) * (x2 - x1) + 15 * l**2 * (x1**2 + x2**2 - 3 * x1 * x2)) / l4
    E41 = 5 * (3 * SQ5 * l * (x2**2 - x1**2) + 3 * l**2 * (x1 + x2) - 10 * x1 * x2 * (x1 + x2)) / l4
    E42 = 5 * (5 * x1**2 + 5 * x2**2 - 3 * l**2 - 3 * SQ5 * l * (x2 - x1) + 20 * x1 * x2) / l4
    E43 = -50 * (X1 + X2) / l4
    E44 = 25 / l4
    E4A41 = E40 + z_m * E41 + (z_m**2 + z_v) * E42 + (z_m**3 + 3 * z_v * z_m) * E43 + (z_m**4 + 6 * z_v * z_m**2 + 3 * z_v**2) * E44
    E4A42 = E41 + (z_m + x1) * E42 + (z_m**2 + 2 * z_v + x1**2 + z_m * x1) * E43 + (z_m**3 + x1**3 + x1 * z_m**2 + z_m * x1**2 + 3 * z_v * x1 + 5 * z_v * z_m) * E44
    E4A43 = E41 + (z_m + x2) * E42 + (z_m**2 + 2 * z_v + x2**2 + z_m * x2) * E43 + (z_m**3 + x2**3 + x2 * z_m**2 + z_m * x2**2 + 3 * z_v * x2 + 5 * z_v * z_m) * E44
    P2 = np.exp(-SQ5 * (x2 - x1) / l) * (
        0.5 * E4A41 * (erf((x2 - z_m) / np.sqrt(2 * z_v)) - erf((x1 - z_m) / np.sqrt(2 * z_v)))
        + E4A42 * np.sqrt(0.5 * z_v / pi) * np.exp(-0.5 * (x1 - z_m)**2 / z_v)
        - E4A43 * np.sqrt(0.5 * z_v / pi) * np.exp(-0.5 * (x2 - z_m)**2 / z_v))

    E50 = 1 + (25 * x1**2 * x2**2 + 3 * SQ5 * (3 * l**3 + 5 * l * x1 * x2) * (x1 + x2) + 15 * l**2 * (x1**2 + x2**2 + 3 * x1 * x2)) / l4
    E51 = (18 * SQ5 * l**3 + 15 * SQ5 * l * (x1**2 + x2**2) + (75 * l**2 + 50 * x1 * x2) * (x1 + x2) + 60 * SQ5 * l * x1 * x2) / l4
    E52 = 5 * (5 * x1**2 + 5 * x2**2 + 15 * l**2 + 9 * SQ5 * l * (x1 + x2) + 20 * x1 * x2) / l4
    E53 = 10 * (3 * SQ5 * l + 5 * x1 + 5 * x2) / l4
    E54 = 25 / l4
    muD = z_m + 2 * SQ5 * z_v / l
    E5A51 = E50 - muD * E51 + (muD**2 + z_v) * E52 - (muD**3 + 3 * z_v * muD) * E53 + (muD**4 + 6 * z_v * muD**2 + 3 * z_v**2) * E54
    E5A52 = E51 - (muD + x1) * E52 + (muD**2 + 2 * z_v + x1**2 + muD * x1) * E53 - (muD**3 + x1**3 + x1 * muD**2 + muD * x1**2 + 3 * z_v * x1 + 5 * z_v * muD) * E54
    P3 = np.exp((10 * z_v - SQ5 * l * (x1 + x2 - 2 * z_m)) / l**2) * (
        0.5 * E5A51 * (1 + erf((x1 - muD) / np.sqrt(2 * z_v)))
        + E5A52 * np.sqrt(0.5 * z_v / pi) * np.exp(-0.5 * (x1 - muD)**2 / z_v))
    return P1 + P2 + P3


def Jd0(x1, z_m, z_v, ell):
    """Diagonal Matern-2.5 J factor (vecchia.py:961-988)."""
    x1 = np.asarray(x1, float)
    l = ell
    l4 = 9 * l ** 4
    pi = np.pi
    E30 = 1 + (25 * x1**4 - 6 * SQ5 * (3 * l**3 + 5 * l * x1**2) * x1 + 75 * l**2 * (x1**2)) / l4
    E31 = (18 * SQ5 * l**3 + 90 * SQ5 * l * x1**2 - (150 * l**2 + 100 * x1**2) * x1) / l4
    E32 = 5 * (30 * x1**2 + 15 * l**2 - 18 * SQ5 * l * x1) / l4
    E33 = 10 * (3 * SQ5 * l - 10 * x1) / l4
    E34 = 25 / l4
    muC = z_m - 2 * SQ5 * z_v / l
    E3A31 = E30 + muC * E31 + (muC**2 + z_v) * E32 + (muC**3 + 3 * z_v * muC) * E33 + (muC**4 + 6 * z_v * muC**2 + 3 * z_v**2) * E34
    E3A32 = E31 + (muC + x1) * E32 + (muC**2 + 2 * z_v + x1**2 + muC * x1) * E33 + (muC**3 + x1**3 + x1 * muC**2 + muC * x1**2 + 3 * z_v * x1 + 5 * z_v * muC) * E34
    P1 = np.exp((10 * z_v + SQ5 * l * (2 * x1 - 2 * z_m)) / l**2) * (
        0.5 * E3A31 * (1 + erf((muC - x1) / np.sqrt(2 * z_v)))
        + E3A32 * np.sqrt(0.5 * z_v / pi) * np.exp(-0.5 * (x1 - muC)**2 / z_v))
    E50 = 1 + (25 * x1**4 + 6 * SQ5 * (3 * l**3 + 5 * l * x1**2) * x1 + 75 * l**2 * (x1**2)) / l4
    E51 = (18 * SQ5 * l**3 + 90 * SQ5 * l * x1**2 + (150 * l**2 + 100 * x1**2) * x1) / l4
    E52 = 5 * (30 * x1**2 + 15 * l**2 + 18 * SQ5 * l * x1) / l4
    E53 = 10 * (3 * SQ5 * l + 10 * x1) / l4
    E54 = 25 / l4
    muD = z_m + 2 * SQ5 * z_v / l
    E5A51 = E50 - muD * E51 + (muD**2 + z_v) * E52 - (muD**3 + 3 * z_v * muD) * E53 + (muD**4 + 6 * z_v * muD**2 + 3 * z_v**2) * E54
    E5A52 = E51 - (muD + x1) * E52 + (muD**2 + 2 * z_v + x1**2 + muD * x1) * E53 - (muD**3 + x1**3 + x1 * muD**2 + muD * x1**2 + 3 * z_v * x1 + 5 * z_v * muD) * E54
    P3 = np.exp((10 * z_v - SQ5 * l * (2 * x1 - 2 * z_m)) / l**2) * (
        0.5 * E5A51 * (1 + erf((x1 - muD) / np.sqrt(2 * z_v)))
        + E5A52 * np.sqrt(0.5 * z_v / pi) * np.exp(-0.5 * (x1 - muD)**2 / z_v))
    return P1 + P3


def IJ(X, z_m, z_v, length, name):
    """I (n,), J (n,n) for one test point with input N(z_m, diag z_v).

    sexp: direct form of IJ_nb (vecchia.py:845-869) == IJ_sexp (functions.py:432-451);
    matern2.5: IJ_matern (functions.py:453-494) with the v_k = 0 branches."""
    X = np.asarray(X, float)
    n, d = X.shape
    length = np.asarray(length, float)
    if len(length) == 1:
        length = np.full(d, length[0])
    if name == 'sexp':
        Xz = X - z_m
        I1 = 1.0 / np.sqrt(np.prod(1 + 2 * z_v / length**2))
        J1 = 1.0 / np.sqrt(np.prod(1 + 4 * z_v / length**2))
        I = I1 * np.exp(-np.sum(Xz**2 / (2 * z_v + length**2), axis=1))
        e = np.zeros((n, n))
        for k in range(d):
            a = Xz[:, k]
            e += (a[:, None] + a[None, :])**2 / (8 * z_v[k] + 2 * length[k]**2) + (a[:, None] - a[None, :])**2 / (2 * length[k]**2)
        J = J1 * np.exp(-e)
        return I, J
    I = np.ones(n)
    J = np.ones((n, n))
    eye = np.eye(n, dtype=bool)
    for k in range(d):
        xk = X[:, k]
        Ik = matern_I_dim(xk, z_m[k], z_v[k], length[k])
        I *= Ik
        if z_v[k] != 0:
            with np.errstate(all='ignore'):
                Jk = Jd(xk[None, :], xk[:, None], z_m[k], z_v[k], length[k])
            Jk[eye] = Jd0(xk, z_m[k], z_v[k], length[k])
        else:
            p = _matern_point(z_m[k] - xk, length[k])
            Jk = p[:, None] * p[None, :]
        J *= Jk
    return I, J


def IJ_sexp_gemm(X, z_m, z_v, length):
    """The SExp I, J of IJ() with the pair exponent expanded: sum_k c1_k (a_i + a_j)^2 + c2_k (a_i - a_j)^2
    = r_i + r_j + sum_k 2 (c1_k - c2_k) a_ik a_jk, r_i = sum_k (c1_k + c2_k) a_ik^2 -- one BLAS product instead of 2 d passes over
    n x n arrays.  Same numbers to rounding (tests/test_oracle_golden.py pins it on IJ); lets the full-size GPU tests compare
    tens of test points at n = 5000 within a minute."""
    X = np.asarray(X, float)
    n, d = X.shape
    length = np.asarray(length, float)
    if len(length) == 1:
        length = np.full(d, length[0])
    a = X - z_m
    c1 = 1.0 / (8 * z_v + 2 * length**2)
    c2 = 1.0 / (2 * length**2)
    I = np.exp(-np.sum(a**2 / (2 * z_v + length**2), axis=1)) / np.sqrt(np.prod(1 + 2 * z_v / length**2))
    r = (a**2) @ (c1 + c2)
    e = (a * (2.0 * (c1 - c2))) @ a.T
    e += r[:, None]
    e += r[None, :]
    np.negative(e, out=e)
    np.exp(e, out=e)
    e *= 1.0 / np.sqrt(np.prod(1 + 4 * z_v / length**2))
    return I, e


def link_gp_predict(m, v, z, W, Wg, Rinv, Rinv_y, scale, length, nugget, name, gemm_form=False):
    """functions.link_gp (functions.py:396-430): per test point
    mean = I.Rinv_y ; var = |Rinv_y^T J Rinv_y - mean^2 + scale(1+nugget-tr(Rinv J))|.
    m, v: (M, Dw) moments of the local inputs; z: (M, Dz) or None deterministic
    global inputs; W (n,Dw), Wg (n,Dz) training inputs."""
    m = np.asarray(m, float)
    v = np.asarray(v, float)
    M = m.shape[0]
    Dw = W.shape[1]
    length = np.asarray(length, float)
    Dz = 0 if z is None else z.shape[1]
    if len(length) == 1:
        length = np.full(Dw + Dz, length[0])
    scale = float(np.ravel(scale)[0])
    nugget = float(np.ravel(nugget)[0])
    mo = np.zeros(M)
    vo = np.zeros(M)
    for t in range(M):
        I, J = IJ_sexp_gemm(W, m[t], v[t], length[:Dw]) if (gemm_form and name == 'sexp') else IJ(W, m[t], v[t], length[:Dw], name)
        if z is not None:
            Iz = cross_corr(Wg, z[t:t + 1], length[Dw:], name)[:, 0]
            I = I * Iz
            J = J * np.outer(Iz, Iz)
        tr = np.sum(Rinv * J)
        mu = I @ Rinv_y
        mo[t] = mu
        vo[t] = np.abs(Rinv_y @ J @ Rinv_y - mu**2 + scale * (1 + nugget - tr))
    return mo, vo


def aggregate_moments(mu_list, var_list):
    """emulation.py:846-847: mu = mean_s mu_s ; s2 = mean_s(mu_s^2+v_s) - mu^2."""
    mu_s = np.asarray(mu_list)
    v_s = np.asarray(var_list)
    mu = np.mean(mu_s, axis=0)
    return mu, np.mean(mu_s**2 + v_s, axis=0) - mu**2


# --------------------------------------------------------------------------
# a17  Vecchia neighbour search (vecchia.py:20-109) -- exact brute force
# --------------------------------------------------------------------------
def nn_ordered(x, m):
    """NNarray (n, m+1) int64 of vecchia.nn: row i = i and its <=m nearest
    EARLIER points (index <= i, squared-euclidean, exact), sorted by index
    descending, -1 padded (vecchia.py:108)."""
    x = np.asarray(x, float)
    n = x.shape[0]
    m = min(m, n - 1)
    out = np.full((n, m + 1), -1, dtype=np.int64)
    for i in range(n):
        d = np.sum((x[:i + 1] - x[i])**2, axis=1)
        k = min(m + 1, i + 1)
        idx = np.argsort(d, kind='stable')[:k]
        out[i, :k] = idx
    return np.fliplr(np.sort(out, axis=1))


def pred_nn(query, x, m):
    """get_pred_nn (vecchia.py:20-40): m nearest training points, nearest first."""
    query = np.asarray(query, float)
    x = np.asarray(x, float)
    n = x.shape[0]
    m = min(m, n)
    if m == n:
        k = query.shape[0]
        return (np.arange(m)[None, :] + np.arange(k)[:, None]) % m
    d = ((query[:, None, :] - x[None, :, :])**2).sum(-1)
    return np.argsort(d, axis=1, kind='stable')[:, :m].astype(np.int64)


# --------------------------------------------------------------------------
# a18-a21  Vecchia likelihoods and sampler (vecchia.py:111-242,292-424)
# --------------------------------------------------------------------------
def _row_idx(NNarray, i):
    idx = NNarray[i]
    return idx[idx >= 0][::-1]


def vecchia_llik(X, y, NNarray, scale, length, nugget, nugget_diag, name):
    """vecchia.py:164-180: -0.5 (sum_i 2 log L_i[last,last] + sum_i (L_i^-1 y_i)_last^2 / scale)."""
    X = np.asarray(X, float)
    y = np.asarray(y, float).reshape(-1)
    n = X.shape[0]
    quad = 0.0
    logdet = 0.0
    for i in range(n):
        idx = _row_idx(NNarray, i)
        Ki = corr_matrix(X[idx], length, name)
        b = len(idx)
        Ki[np.arange(b), np.arange(b)] = 1.0 + nugget * nugget_diag[idx]
        Li = np.linalg.cholesky(Ki)
        w = solve_triangular(Li, y[idx], lower=True)
        quad += w[-1]**2
        logdet += 2 * np.log(np.abs(Li[-1, -1]))
    return -0.5 * (logdet + quad / scale)


def vecchia_nllik(X, y, NNarray, scale, length, nugget, nugget_diag, name,
                  scale_est, nugget_est, origin_n, rr):
    """vecchia.py:182-242 -> (nll, grad(p), scale)."""
    X = np.asarray(X, float)
    y = np.asarray(y, float).reshape(-1)
    length = np.asarray(length, float)
    n = X.shape[0]
    p = len(length) + (1 if nugget_est else 0)
    dquad = np.zeros(p)
    dlogdet = np.zeros(p)
    quad = 0.0
    logdet = 0.0
    for i in range(n):
        idx = _row_idx(NNarray, i)
        b = len(idx)
        Ki, dKi = k_matrix_fod(X[idx], length, 0.0, name, False)
        nug = nugget * nugget_diag[idx]
        Ki[np.arange(b), np.arange(b)] = 1.0 + nug
        if nugget_est:
            dKi = np.concatenate((dKi, np.diag(nug)[None]), axis=0)
        Li = np.linalg.cholesky(Ki)
        w = solve_triangular(Li, y[idx], lower=True)
        e = np.zeros(b)
        e[-1] = 1.0
        u = solve_triangular(Li.T, e, lower=False)
        for k in range(p):
            t = solve_triangular(Li, dKi[k] @ u, lower=True)
            s = w @ t
            dquad[k] += 2 * s * w[-1] - t[-1] * w[-1]**2
            dlogdet[k] += t[-1]
        quad += w[-1]**2
        logdet += 2 * np.log(np.abs(Li[-1, -1]))
    if scale_est:
        if n == origin_n:
            scale = quad / n
            nll = 0.5 * (logdet + n * np.log(scale))
            g = 0.5 * (dlogdet - dquad / scale)
        else:
            scale = (quad + rr / nugget) / origin_n
            nll = 0.5 * (logdet + origin_n * np.log(scale))
            g = 0.5 * (dlogdet - dquad / scale)
            if nugget_est:
                nll += 0.5 * (origin_n - n) * np.log(nugget)
                g[-1] += 0.5 * (-rr / (scale * nugget) + (origin_n - n))
    else:
        nll = 0.5 * (logdet + quad / scale)
        g = 0.5 * (dlogdet - dquad / scale)
        if n != origin_n and nugget_est:
            nll += 0.5 * (rr / (nugget * scale) + (origin_n - n) * np.log(nugget))
            g[-1] += 0.5 * (-rr / (scale * nugget) + (origin_n - n))
    return float(nll), g, float(scale)


def L_matrix(X, NNarray, length, nugget, name):
    """vecchia.py:409-424: row i = (e_last^T L_i^-1) reversed (self first), zero padded."""
    X = np.asarray(X, float)
    n, mp1 = NNarray.shape
    out = np.zeros((n, mp1))
    for i in range(n):
        idx = _row_idx(NNarray, i)
        b = len(idx)
        Ki = corr_matrix(X[idx], length, name)
        Ki[np.arange(b), np.arange(b)] = 1.0 + nugget
        Li = np.linalg.cholesky(Ki)
        e = np.zeros(b)
        e[-1] = 1.0
        u = solve_triangular(Li.T, e, lower=False)
        out[i, :b] = u[::-1]
    return out


def forward_solve_sp(L, NNarray, b):
    """vecchia.py:111-120 sequential sparse lower solve."""
    n, m = L.shape
    x = np.zeros(n)
    for i in range(n):
        s = 0.0
        for j in range(1, min(i + 1, m)):
            s += L[i, j] * x[NNarray[i, j]]
        x[i] = (b[i] - s) / L[i, 0]
    return x


def fmvn_sp(X, NNarray, scale, length, nugget, name, z):
    """vecchia.py:133-140 with injected z."""
    L = L_matrix(X, NNarray, length, nugget, name) / np.sqrt(scale)
    return forward_solve_sp(L, NNarray, z)


# --------------------------------------------------------------------------
# a22/a23  Vecchia prediction (vecchia.py:635-654,758-796)
# --------------------------------------------------------------------------
def gp_vecch(x, w, NNarray, y, scale, length, nugget, nugget_diag, name):
    x = np.asarray(x, float)
    w = np.asarray(w, float)
    y = np.asarray(y, float).reshape(-1)
    M = x.shape[0]
    mo = np.zeros(M)
    vo = np.zeros(M)
    for i in range(M):
        idx = NNarray[i]
        idx = idx[idx >= 0]
        Xi = np.vstack((w[idx], x[i:i + 1]))
        b = len(idx) + 1
        Ki = corr_matrix(Xi, length, name)
        nug = np.empty(b)
        nug[:-1] = nugget * nugget_diag[idx]
        nug[-1] = nugget
        Ki[np.arange(b), np.arange(b)] = 1.0 + nug
        Li = np.linalg.cholesky(Ki)
        mo[i] = Li[-1, :-1] @ solve_triangular(Li[:-1, :-1], y[idx], lower=True)
        vo[i] = scale * Li[-1, -1]**2
    return mo, vo


def loo_gp_vecch(x, NNarray, y, scale, length, nugget, nugget_diag, name):
    """vecchia.loo_gp_vecch (vecchia.py:656-674): row i of NNarray = the point itself followed by its nearest
    neighbours; the block is ordered self-last and the point is predicted from the others.  Unlike gp_vecch the
    point's own nugget carries its replicate weight."""
    x = np.asarray(x, float)
    y = np.asarray(y, float).reshape(-1)
    n = x.shape[0]
    mo, vo = np.zeros(n), np.zeros(n)
    for i in range(n):
        idx = NNarray[i]
        idx = idx[idx >= 0][::-1]
        Ki = corr_matrix(x[idx], length, name)
        b = len(idx)
        Ki[np.arange(b), np.arange(b)] = 1.0 + nugget * nugget_diag[idx]
        Li = np.linalg.cholesky(Ki)
        mo[i] = Li[-1, :-1] @ solve_triangular(Li[:-1, :-1], y[idx][:-1], lower=True)
        vo[i] = scale * Li[-1, -1]**2
    return mo, vo


def imp_nn_array(Xs_ord, m):
    """kernel.ord_nn(pointer=True) (kernel_class.py:268-274): per ordered point i the conditioning set of its LATENT
    value in the stacked vector [observations (0..n-1) ; latents (n..2n-1)]: its own latent (n+i), its own
    observation (i) and its m-1 nearest other points -- as latents if they come earlier in the ordering, as
    observations otherwise."""
    Xs_ord = np.asarray(Xs_ord, float)
    n = Xs_ord.shape[0]
    NNs = pred_nn(Xs_ord, Xs_ord, m)[:, 1:].copy()
    prev = NNs < np.arange(n)[:, None]
    NNs[prev] += n
    return np.hstack((np.arange(n).reshape(-1, 1) + n, np.arange(n).reshape(-1, 1), NNs)).astype(np.int64)


def U_matrix_rows(X_ord, impNN, scale, length, name, gamma2):
    """vecchia.U_matrix through U_matrix_sp's argument conventions (vecchia.py:426-446,599-610): row i = last column of
    L_i^-T for the block [.., own observation, own latent] of scale*corr + diag(gamma on observation entries + 1e-10)."""
    X_ord = np.asarray(X_ord, float)
    n = X_ord.shape[0]
    X2 = np.vstack((X_ord, X_ord))
    rev = impNN[:, ::-1]
    U = np.zeros(rev.shape)
    for i in range(n):
        idx = rev[i]
        idx = idx[idx >= 0]
        cond = idx > n - 1
        b = len(idx)
        Ki = scale * corr_matrix(X2[idx], length, name)
        Ki[np.arange(b), np.arange(b)] = scale * 1.0 + gamma2[idx] * ~cond + 1e-10
        Li = np.linalg.cholesky(Ki)
        e = np.zeros(b)
        e[-1] = 1.0
        U[i, :b] = solve_triangular(Li.T, e, lower=False)
    return U


def post_het_vecch(X_ord, impNN, scale, length, name, gamma2, y_ord, z):
    """Hetero.post_het_vecch on U_matrix_sp's output (likelihood_class.py:166-182, vecchia.py:599-610): with U (2n x n)
    assembled from the rows, U_l its latent block and U_ol its observation block,
    f = -U_l^-T U_ol^T y + U_l^-T z   (ordered coordinates)."""
    n = len(y_ord)
    rows = U_matrix_rows(X_ord, impNN, scale, length, name, gamma2)
    rev = impNN[:, ::-1]
    U = np.zeros((2 * n, n))
    for i in range(n):
        idx = rev[i]
        keep = idx >= 0
        U[idx[keep], i] = rows[i, :keep.sum()]
    U_l, U_ol = U[n:], U[:n]
    L = U_l.T
    mu = -solve_triangular(L, U_ol.T @ np.asarray(y_ord, float).reshape(-1), lower=True)
    return mu + solve_triangular(L, np.asarray(z, float).reshape(-1), lower=True)


def link_gp_vecch(m, v, z, w1, global_w1, NNarray, y, scale, length, nugget, nugget_diag, name):
    m = np.asarray(m, float)
    v = np.asarray(v, float)
    y = np.asarray(y, float).reshape(-1)
    M = m.shape[0]
    Dw = w1.shape[1]
    Dz = 0 if z is None else z.shape[1]
    length = np.asarray(length, float)
    if len(length) == 1:
        length = np.full(Dw + Dz, length[0])
    mo = np.zeros(M)
    vo = np.zeros(M)
    for i in range(M):
        idx = NNarray[i]
        idx = idx[idx >= 0]
        b = len(idx)
        wi = w1[idx]
        I, J = IJ(wi, m[i], v[i], length[:Dw], name)
        if z is not None:
            gi = global_w1[idx]
            Iz = cross_corr(gi, z[i:i + 1], length[Dw:], name)[:, 0]
            I = I * Iz
            J = J * np.outer(Iz, Iz)
            Ki = corr_matrix(np.concatenate((wi, gi), 1), length, name)
        else:
            Ki = corr_matrix(wi, length, name)
        Ki[np.arange(b), np.arange(b)] = 1.0 + nugget * nugget_diag[idx]
        tr = np.trace(np.linalg.solve(Ki, J))
        Li = np.linalg.cholesky(Ki)
        Ry = cho_solve((Li, True), y[idx])
        mu = I @ Ry
        mo[i] = mu
        vo[i] = np.abs(Ry @ J @ Ry - mu**2 + scale * (1 + nugget - tr))
    return mo, vo


def mice_var(x, x_extra, input_dim, connect, name, length, scale, nugget, nugget_s):
    """functions.mice_var (functions.py:244-256): smoothed predictive variance of a GP whose design is the candidate
    set itself, scale / diag(pinvh(R)) with R built on [x[:, input_dim] | x_extra[:, connect]] and the nugget
    max(nugget_s, nugget)."""
    from scipy.linalg import pinvh
    Xin = x[:, input_dim]
    if connect is not None:
        Xin = np.concatenate((Xin, x_extra[:, connect]), 1)
    R = k_matrix(Xin, np.asarray(length, float), max(nugget_s, nugget), name)
    return (float(np.asarray(scale).reshape(-1)[0]) / np.diag(pinvh(R, check_finite=False))).reshape(-1, 1)


# ------------------------------------------------------------------ heteroskedastic Gaussian likelihood
def hetero_llik(inp, out):
    """Hetero.llik (likelihood_class.py:108-113): sum_i -0.5 (log 2pi + logvar_i + (y_i - mu_i)^2 / exp(logvar_i)),
    the ratio formed as exp(log r2 - logvar) like the reference."""
    mu, log_var = inp[:, 0], inp[:, 1]
    r2 = (np.asarray(out).flatten() - mu) ** 2
    with np.errstate(divide='ignore'):
        return float(np.sum(-0.5 * (np.log(2 * np.pi) + log_var + np.exp(np.log(r2) - log_var))))


def hetero_prediction(m, v):
    """Hetero.prediction (likelihood_class.py:123-127): mean m_0, variance exp(m_1 + v_1 / 2) + v_0."""
    return m[:, 0].flatten(), (np.exp(m[:, 1] + v[:, 1] / 2) + v[:, 0]).flatten()


def post_het1(v, Gamma, y, sd):
    """Hetero.post_het1 (likelihood_class.py:184-207) with the normals sd (n x 2) injected: a draw of the mean latent
    from its exact conditional posterior, f = mu + u - v (v + diag Gamma)^-1 (u + w), mu = v (v + diag Gamma)^-1 y,
    u = chol(v) sd[:,0], w = sqrt(Gamma) sd[:,1]."""
    vG = v + np.diag(Gamma)
    c = cho_factor(vG, lower=True, check_finite=False)
    L1 = np.linalg.cholesky(v)
    mu = v @ cho_solve(c, np.asarray(y).flatten(), check_finite=False)
    u = L1 @ sd[:, 0]
    w = np.sqrt(Gamma) * sd[:, 1]
    return -v @ cho_solve(c, u + w, check_finite=False) + mu + u


def post_het2(v, Gamma, mask, y, sd):
    """Hetero.post_het2 (likelihood_class.py:209-243): the same with replicates -- observation i belongs to site
    mask[i]; per site the precision-weighted sums M Gamma^-1 y and M Gamma^-1 M replace y and Gamma."""
    N = v.shape[0]
    Gi = 1.0 / Gamma
    MGy = np.bincount(mask, weights=Gi * np.asarray(y).flatten(), minlength=N)
    iMGM = 1.0 / np.bincount(mask, weights=Gi, minlength=N)
    c = cho_factor(v + np.diag(iMGM), lower=True, check_finite=False)
    L1 = np.linalg.cholesky(v)
    mu = v @ cho_solve(c, iMGM * MGy, check_finite=False)
    u = L1 @ sd[:, 0]
    w = np.sqrt(iMGM) * sd[:, 1]
    return -v @ cho_solve(c, u + w, check_finite=False) + mu + u


def hetero_pllik(y, f):
    """Hetero.pllik (likelihood_class.py:115-121): pointwise Gaussian log-density with mean f[..., 0] and
    log-variance f[..., 1]; y (M, 1, 1)-broadcastable, f (M, Q, 2) -> (M, Q, 1)."""
    mu, var = f[:, :, [0]], np.exp(f[:, :, [1]])
    return -0.5 * (np.log(2 * np.pi * var) + (y - mu) ** 2 / var)


def ghdiag(fct, mu, var, y):
    """functions.ghdiag (functions.py:233-241): E[exp(fct(y, f))] under f ~ N(mu, diag(var)) by the tensor-product
    10-point Gauss-Hermite rule (mu, var: M x N latents per test point) -> (M, 1)."""
    import itertools
    x, w = np.polynomial.hermite.hermgauss(10)
    N = mu.shape[1]
    xn = np.array(list(itertools.product(*(x,) * N)))
    wn = np.prod(np.array(list(itertools.product(*(w,) * N))), 1)[:, None]
    fn = np.sqrt(2.0) * (np.sqrt(var[:, None]) * xn) + mu[:, None]
    ll = fct(y[:, None], fn)
    return np.sum(np.exp(np.log((wn * np.pi ** (-0.5 * N))[None, :]) + ll), axis=1)


# --------------------------------------------------------------------------
# count likelihoods (likelihood_class.py:8-90 Poisson, :245-292 NegBin) and their latent warm starts (dgp.py:327-336,526-566)
# --------------------------------------------------------------------------
def poisson_pllik(y, f):
    from scipy.special import gammaln
    return y * f - np.exp(f) - gammaln(y + 1)


def poisson_prediction(m, v):
    mean = np.exp(m + v / 2)
    return mean.flatten(), (mean + (np.exp(v) - 1) * np.exp(2 * m + v)).flatten()


def negbin_pllik(y, f):
    from scipy.special import gammaln
    f1, f2 = f[..., [0]], f[..., [1]]
    size, a = np.exp(-f2), f1 + f2
    return gammaln(y + size) - gammaln(size) - gammaln(y + 1.0) + y * a - (y + size) * np.logaddexp(0.0, a)


def negbin_prediction(m, v):
    e_mu = np.exp(m[:, 0] + v[:, 0] / 2)
    var = (np.exp(2 * m[:, 0] + v[:, 0]) * (np.exp(v[:, 0]) - 1) + e_mu
           + np.exp(m[:, 1] + v[:, 1] / 2) * np.exp(2 * m[:, 0] + 2 * v[:, 0]))
    return e_mu, var


def zip_pllik(y, f):
    """ZIP log-pmf (likelihood_class.py:497-572): zero = log(pi + (1-pi) e^-lam), y>0 = log(1-pi) - lam + y f - log y!."""
    from scipy.special import gammaln, expit
    f_lam, f_pi = f[..., [0]], f[..., [1]]
    lam, pi = np.exp(f_lam), expit(f_pi)
    y = np.broadcast_to(y, lam.shape)
    out = np.empty(lam.shape)
    z = y == 0
    out[z] = np.logaddexp(np.log(pi[z]), np.log1p(-pi[z]) - lam[z])
    out[~z] = np.log1p(-pi[~z]) - lam[~z] + y[~z] * f_lam[~z] - gammaln(y[~z] + 1.0)
    return out


def zip_prediction(m, v):
    from scipy.special import expit
    lam_mean = np.exp(m[:, 0] + 0.5 * v[:, 0])
    lam_var = (np.exp(v[:, 0]) - 1.0) * np.exp(2.0 * m[:, 0] + v[:, 0])
    den = np.maximum(1.0 + (np.pi / 8.0) * v[:, 1], 1e-12)
    p = expit(m[:, 1] / np.sqrt(den))
    p_var = np.clip((p * (1.0 - p))**2 * v[:, 1] / den, 0.0, p * (1.0 - p))
    mean = (1.0 - p) * lam_mean
    var = (1.0 - p) * lam_mean * (1.0 + p * lam_mean) + ((1.0 - p)**2 + p_var) * lam_var + p_var * lam_mean**2
    return mean, np.maximum(var, 0.0)


def zip_warm_start(y, rep, G):
    """dgp.py:337-410."""
    out = np.empty((G, 2))
    lam_floor, pi_min, pi_max = 1e-6, 1e-4, 0.99
    if rep is None:
        out[:, 0] = np.log(np.maximum(y + 0.5, lam_floor) + 1e-12)
        p0 = ((y == 0).sum() + 0.5) / (len(y) + 1.0)
        mu = y.mean()
        if mu <= 0:
            pi0 = p0
        else:
            q0 = np.exp(-max(mu, lam_floor))
            pi0 = 0.0 if q0 >= 1.0 - 1e-8 else np.clip((p0 - q0) / (1.0 - q0), 0.0, pi_max)
        pi0 = np.clip(pi0, pi_min, 1.0 - pi_min)
        out[:, 1] = np.log(pi0 / (1.0 - pi0))
        return out
    n_g = np.bincount(rep, minlength=G)
    mu_g = np.bincount(rep, weights=y, minlength=G) / np.maximum(n_g, 1)
    p0_g = (np.bincount(rep, weights=(y == 0).astype(float), minlength=G) + 0.1) / (n_g + 0.2)
    lam0 = mu_g.copy()
    lam0[mu_g == 0.0] = y[y > 0].mean() if np.any(y > 0) else 1.0
    lam0 = np.maximum(lam0, lam_floor)
    q = np.exp(-lam0)
    pi_g = np.clip(np.where(p0_g <= q, 0.0, (p0_g - q) / np.maximum(1.0 - q, 1e-8)), 0.0, pi_max)
    lam = np.maximum(np.where(mu_g == 0.0, lam0, mu_g / np.maximum(1.0 - pi_g, 1e-3)), lam_floor)
    pi_g = np.clip(pi_g, pi_min, 1.0 - pi_min)
    out[:, 0] = np.log(lam + 1e-12)
    out[:, 1] = np.log(pi_g / (1.0 - pi_g))
    return out


def zinb_pllik(y, f):
    """ZINB log-pmf (likelihood_class.py:653-739)."""
    from scipy.special import expit
    nb = negbin_pllik(y, f[..., :2])
    pi = expit(f[..., 2:3])
    yb = np.broadcast_to(y, nb.shape)
    return np.where(yb == 0, np.logaddexp(np.log(pi), np.log1p(-pi) + nb), np.log1p(-pi) + nb)


def zinb_prediction(m, v):
    from scipy.special import expit
    mu_mean = np.exp(m[:, 0] + 0.5 * v[:, 0])
    mu_var = (np.exp(v[:, 0]) - 1.0) * np.exp(2.0 * m[:, 0] + v[:, 0])
    mu2 = np.exp(2.0 * m[:, 0] + 2.0 * v[:, 0])
    mu2_sig = mu2 * np.exp(m[:, 1] + 0.5 * v[:, 1])
    den = np.maximum(1.0 + (np.pi / 8.0) * v[:, 2], 1e-12)
    p = expit(m[:, 2] / np.sqrt(den))
    p_var = np.clip((p * (1.0 - p))**2 * v[:, 2] / den, 0.0, p * (1.0 - p))
    e_p1m = np.clip(p * (1.0 - p) - p_var, 0.0, p * (1.0 - p))
    var = (1.0 - p) * (mu_mean + mu2_sig) + e_p1m * mu2 + ((1.0 - p)**2 + p_var) * mu_var + p_var * mu_mean**2
    return (1.0 - p) * mu_mean, np.maximum(var, 0.0)


def zinb_warm_start(y, rep, G):
    """dgp.py:411-525."""
    out = np.empty((G, 3))
    lam_floor, pi_min, pi_max, eps = 1e-6, 1e-4, 0.99, 1e-8
    y_mean = y.mean()
    sg = (y.var(ddof=1) - y_mean) / (y_mean**2 + eps) if y.size > 1 else 1.0
    sg = min(max(sg, 1e-3), 10.0)
    if rep is None:
        out[:, 0] = np.log(np.maximum(y + 0.5, lam_floor) + 1e-12)
        out[:, 1] = np.log(sg)
        p0 = ((y == 0).sum() + 0.5) / (len(y) + 1.0)
        if y_mean <= 0:
            pi0 = p0
        else:
            q0 = np.exp(-max(y_mean, lam_floor))
            pi0 = 0.0 if q0 >= 1.0 - 1e-8 else np.clip((p0 - q0) / (1.0 - q0), 0.0, pi_max)
        pi0 = np.clip(pi0, pi_min, 1.0 - pi_min)
        out[:, 2] = np.log(pi0 / (1.0 - pi0))
        return out
    cnt = np.bincount(rep, minlength=G).astype(float)
    s1 = np.bincount(rep, weights=y, minlength=G)
    s2 = np.bincount(rep, weights=y * y, minlength=G)
    mu_g = (s1 + 0.5) / np.maximum(cnt, 1.0)
    out[:, 0] = np.log(mu_g + 1e-12)
    var_hat = mu_g.copy()
    mk = cnt > 1
    var_hat[mk] = (s2[mk] - s1[mk]**2 / cnt[mk]) / (cnt[mk] - 1.0)
    sig = (var_hat - mu_g) / (mu_g**2 + eps)
    sig[(~np.isfinite(sig)) | (sig <= 0.0)] = sg
    out[:, 1] = np.log(np.clip(sig, 1e-3, 10.0))
    p0_g = (np.bincount(rep, weights=(y == 0).astype(float), minlength=G) + 0.1) / (cnt + 0.2)
    mu_raw = s1 / np.maximum(cnt, 1.0)
    lam0 = mu_raw.copy()
    lam0[mu_raw == 0.0] = y[y > 0].mean() if np.any(y > 0) else 1.0
    q = np.exp(-np.maximum(lam0, lam_floor))
    pi_g = np.clip(np.where(p0_g <= q, 0.0, (p0_g - q) / np.maximum(1.0 - q, 1e-8)), 0.0, pi_max)
    pi_g = np.clip(pi_g, pi_min, 1.0 - pi_min)
    out[:, 2] = np.log(pi_g / (1.0 - pi_g))
    return out


def count_warm_start(name, X, Y):
    """Latents fed to a Poisson / NegBin likelihood at initialisation; X with possibly repeated rows.  Returns
    (latent (G x q), rep or None).  NegBin without replicates: only column 0 is defined by the reference."""
    X0, inv = np.unique(X, return_inverse=True, axis=0)
    inv = np.asarray(inv).reshape(-1)
    y = np.asarray(Y, float).ravel()
    rep = None if len(X0) == len(X) else inv
    G = len(X0)
    if name == 'ZIP':
        return zip_warm_start(y, rep, G), rep
    if name == 'ZINB':
        return zinb_warm_start(y, rep, G), rep
    if name == 'Poisson':
        if rep is None:
            return np.log(y + .5 + 1e-12)[:, None], None
        cnt = np.bincount(rep, minlength=G)
        return np.log((np.bincount(rep, weights=y, minlength=G) + .5) / cnt + 1e-12)[:, None], rep
    out = np.full((G, 2), np.nan)
    if rep is None:
        out[:, 0] = np.log(y + .5 + 1e-12)
        return out, None
    eps = 1e-8
    sig_glob = max((y.var(ddof=1) - y.mean()) / (y.mean() ** 2 + eps), 1e-3)
    cnt = np.bincount(rep, minlength=G).astype(float)
    s1 = np.bincount(rep, weights=y, minlength=G)
    s2 = np.bincount(rep, weights=y * y, minlength=G)
    mu = (s1 + .5) / cnt
    out[:, 0] = np.log(mu + 1e-12)
    var_hat = mu.copy()
    mk = cnt > 1
    var_hat[mk] = (s2[mk] - s1[mk] ** 2 / cnt[mk]) / (cnt[mk] - 1.0)
    sig = (var_hat - mu) / (mu ** 2 + eps)
    sig[(~np.isfinite(sig)) | (sig <= 0.0)] = sig_glob
    out[:, 1] = np.log(np.clip(sig, 1e-3, 10.0))
    return out, rep
