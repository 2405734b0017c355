"""M-step tail, device side: run the bench model (configs[1]: n=2000, d=5, Matern-2.5) and dump, for a few
SI iterations, what every node's L-BFGS-B run saw -- inputs, outputs, start point, bounds, options and the sequence of
(x, nll, gradient) the device's objective returned -- so that tools/cpu_mstep_tail_replay.py can run the SAME fits on the CPU
with the oracle's objective under scipy.optimize.minimize and compare the evaluation counts.  Also: the device objective's
evaluation-to-evaluation noise at each node's final point (20 evaluations at x (1 +- j 1e-13)).
usage (GPU box): python tools/gpu_mstep_tail_dump.py [iterations=6] [warmup=5]  ->  gpurun_out/r4_mstep_tail/dump.npz"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dgp_amd import mstep as M

its = int(sys.argv[1]) if len(sys.argv) > 1 else 6
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
model, X, Y = bench.build_model(2000, 5, 100, 0)
for _ in range(warm):
    model.imp.sample(burnin=10)
    model._m_step()

out = {}
log = []
orig = M.minimize_lockstep


def logged(problems, evaluate, on_finish=None):
    def ev(req):
        res = evaluate(req)
        for (i, x), (f, g) in zip(req, res):
            log.append((i, np.array(x, float), float(np.asarray(f).reshape(-1)[0]), np.array(g, float)))
        return res
    return orig(problems, ev, on_finish)


M.minimize_lockstep = logged
nodes = [nd for layer in model.all_layer for nd in layer if nd.type == 'gp']
for it in range(its):
    model.imp.sample(burnin=10)
    for j, nd in enumerate(nodes):   # what the fit of node j starts from (kernel_class.py:516-560)
        x0, lb, ub, opts = nd._opt_setup()
        out['it%d_n%d_X' % (it, j)] = np.array(nd._X(), float)
        out['it%d_n%d_y' % (it, j)] = np.array(nd.output, float)
        out['it%d_n%d_x0' % (it, j)] = np.array(x0, float)
        out['it%d_n%d_lb' % (it, j)] = np.array(lb if lb is not None else [np.nan])
        out['it%d_n%d_ub' % (it, j)] = np.array(ub if ub is not None else [np.nan])
        out['it%d_n%d_meta' % (it, j)] = np.array([float(nd.scale[0]), float(nd.nugget[0]), float(nd.nugget_est), float(nd.scale_est),
                                                  float(opts.get('maxiter', 15000)), float(opts.get('maxfun', 15000)), float(nd.name == 'matern2.5')])
    log.clear()
    model._m_step()
    for j in range(len(nodes)):
        seq = [(x, f, g) for i, x, f, g in log if i == j]
        out['it%d_n%d_xs' % (it, j)] = np.stack([s[0] for s in seq])
        out['it%d_n%d_fs' % (it, j)] = np.array([s[1] for s in seq])
        out['it%d_n%d_gs' % (it, j)] = np.stack([s[2] for s in seq])
    print('iteration %d: device evaluations per node %s' % (it, [len(out['it%d_n%d_fs' % (it, j)]) for j in range(len(nodes))]), flush=True)
    # evaluation-to-evaluation noise of the device objective at every node's final point
    for j, nd in enumerate(nodes):
        xf = out['it%d_n%d_xs' % (it, j)][-1]
        nd._stage()
        nd._in_maximise = True
        fs, gs = [], []
        for r in range(20):
            f, g = nd.llik(xf * (1.0 + (r - 10) * 1e-13))
            fs.append(float(np.asarray(f).reshape(-1)[0]))
            gs.append(np.array(g, float))
        nd._in_maximise = False
        nd.update(xf)
        out['it%d_n%d_noise_f' % (it, j)] = np.array(fs)
        out['it%d_n%d_noise_g' % (it, j)] = np.stack(gs)
    print('   noise at the final points: spread of nll %s' % ['%.1e' % np.ptp(out['it%d_n%d_noise_f' % (it, j)]) for j in range(len(nodes))], flush=True)
out['prior'] = np.array([nodes[0].prior_name == 'ga', *np.asarray(nodes[0].prior_coef, float).ravel()], float)
os.makedirs('gpurun_out/r4_mstep_tail', exist_ok=True)
np.savez_compressed('gpurun_out/r4_mstep_tail/dump.npz', **out)
print('saved', sum(v.nbytes for v in out.values()) / 1e6, 'MB')
