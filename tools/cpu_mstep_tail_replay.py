"""M-step tail, CPU side.  Replays the fits tools/gpu_mstep_tail_dump.py recorded on the GPU box with the
ORACLE's objective (oracle.nll_grad: numpy + LAPACK, the reference's formulation, kernel_class.py:403-449) under
scipy.optimize.minimize(method='L-BFGS-B') with the reference's options (kernel_class.py:516-545: maxiter 100,
maxfun = max(30, 20 + 5 D)), and prints per fit the evaluation counts side by side, the difference between the device's and
the oracle's objective / gradient at the device's own iterates, and both objectives' evaluation-to-evaluation noise.
usage (this container): python tools/cpu_mstep_tail_replay.py gpurun_out/r4_mstep_tail/dump.npz [max_iterations] > profiles/r04_mstep_tail.txt"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.optimize import minimize, Bounds
from oracle import dgp_oracle as O

d = np.load(sys.argv[1])
max_it = int(sys.argv[2]) if len(sys.argv) > 2 else 99
its = sorted({int(k.split('_')[0][2:]) for k in d.files if k.startswith('it')})[:max_it]
nodes = sorted({int(k.split('_')[1][1:]) for k in d.files if k.startswith('it')})
prior = d['prior']
prior_name, prior_coef = ('ga' if prior[0] else 'inv_ga'), prior[1:]
print('fit = (SI iteration, node); node 0-4: first layer (D = 5 inputs), node 5: second layer (D = 10)')
print('%-8s | evaluations: device  oracle | why the oracle run stopped | final nll: device, oracle | max |nll_dev - nll_orc| and rel |g_dev - g_orc| at the device iterates | noise (20 evaluations at x(1 +- j 1e-13)): nll spread device, oracle; |g| spread device, oracle' % 'fit')
hist_dev, hist_orc = [], []
t00 = time.time()
for it in its:
    for j in nodes:
        key = 'it%d_n%d_' % (it, j)
        X, y, x0 = d[key + 'X'], d[key + 'y'], d[key + 'x0']
        scale, nugget, nugget_est, scale_est, maxiter, maxfun, matern = d[key + 'meta']
        name = 'matern2.5' if matern else 'sexp'
        lb, ub = d[key + 'lb'], d[key + 'ub']

        def fun(x):
            f, g, _ = O.nll_grad(x, X, y, name, scale, nugget, bool(nugget_est), bool(scale_est), prior_name, prior_coef)
            return float(np.ravel(f)[0]), np.asarray(g, float)
        kw = dict(method='L-BFGS-B', jac=True, options=dict(maxiter=int(maxiter), maxfun=int(maxfun)))
        if not np.isnan(lb).all():
            kw['bounds'] = Bounds(lb, ub)
        res = minimize(fun, x0, **kw)
        xs, fs, gs = d[key + 'xs'], d[key + 'fs'], d[key + 'gs']
        # the oracle at (a sample of) the device's own iterates
        pick = sorted(set([0, len(xs) // 2, len(xs) - 1]))
        df, dg = 0.0, 0.0
        for i in pick:
            fo, go = fun(xs[i])
            df = max(df, abs(fo - fs[i]))
            dg = max(dg, np.abs(go - gs[i]).max() / max(np.abs(go).max(), 1e-300))
        # noise of the oracle at the device's final point, same perturbations
        xf = xs[-1]
        nf, ng = [], []
        for r in range(0, 20, 4):
            fo, go = fun(xf * (1.0 + (r - 10) * 1e-13))
            nf.append(fo); ng.append(go)
        nfd, ngd = d[key + 'noise_f'], d[key + 'noise_g']
        hist_dev.append(len(fs)); hist_orc.append(res.nfev)
        print('(%2d, %d) | %6d %7d | %-28s | %.9e %.9e | %.1e %.1e | %.1e %.1e ; %.1e %.1e' % (
            it, j, len(fs), res.nfev, str(res.message)[:28], fs[-1], res.fun, df, dg, np.ptp(nfd), np.ptp(nf),
            np.ptp(ngd, axis=0).max(), np.ptp(np.stack(ng), axis=0).max()), flush=True)
print('evaluations per fit, device:', sorted(hist_dev))
print('evaluations per fit, oracle:', sorted(hist_orc))
print('slowest node per iteration (= lock-step rounds), device: %s' % [max(hist_dev[i:i + len(nodes)]) for i in range(0, len(hist_dev), len(nodes))])
print('slowest node per iteration, oracle: %s' % [max(hist_orc[i:i + len(nodes)]) for i in range(0, len(hist_orc), len(nodes))])
print('total evaluations: device %d, oracle %d   (%.0f s of CPU)' % (sum(hist_dev), sum(hist_orc), time.time() - t00))
