"""I-step / M-step wall time of the bench's SI iteration under the factorisation modes and with / without the
device-queued ESS loop.  usage: gpu_iter_split.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model

model, X, Y = build_model(2000, 5, 100, 0)
imp = model.imp
eng = model.engine
for mode in (0, 2, 1):
    for queued in (False, True):
        eng.set_potrf_mode(mode)
        imp.queued = queued
        for w_ in range(3):
            try:
                if os.environ.get('PRESYNC'):
                    torch.cuda.synchronize()
                imp.sample(burnin=10)
            except Exception as ex:
                print('FAILED in warm-up', w_, 'of', (mode, queued), repr(ex), imp.stats)
                try:
                    sigs, buf = imp._factor_cache[0]
                    cached = buf.clone()
                    n_ = imp.F[0].shape[0]
                    eng.set_potrf_mode(0)
                    imp._factor_cache = {}
                    fresh = imp._layer_factors(0, list(range(5))).clone()
                    for j in range(5):
                        dl = (torch.tril(cached[j][:n_, :n_]) - torch.tril(fresh[j][:n_, :n_])).abs().max().item()
                        print('   first-layer factor %d: |cached - recomputed (mode 0)| = %.3e, NaNs in cached %d' % (j, dl, int(torch.isnan(torch.tril(cached[j][:n_, :n_])).sum())))
                    eng.set_potrf_mode(mode)
                    imp._factor_cache = {0: (sigs, cached)}
                except Exception as ex3:
                    print('   factor comparison failed', repr(ex3))
                try:
                    imp._attach()
                    nu_ = imp._prior_draws_ahead(2)
                    F_ = imp.F[0]
                    print('   F: max|.| %.3e NaN %d | nu: max|.| %.3e NaN %d' % (F_.abs().max().item(), int(torch.isnan(F_).sum()), nu_.abs().max().item(), int(torch.isnan(nu_).sum())))
                    for md in (0, 1):
                        eng.set_potrf_mode(md)
                        ll_, info_ = imp._upper_loglik(0, F_[None])
                        FP_ = eng.ess_propose(F_, nu_[0], [0.3, -0.2, 1.0])
                        ll2_, info2_ = imp._upper_loglik(0, FP_)
                        print('   mode %d: current state ll %s info %s | three proposals ll %s info %s' % (md, ll_, info_, ll2_, info2_))
                    nd2 = model.all_layer[1][0]
                    print('   upper node: length', nd2.length, 'nugget', nd2.nugget, 'scale', nd2.scale)
                    # the failing call's own buffers: 12 proposals through the plan's A / workspace
                    plan = imp._ess_plan(0)
                    n_ = F_.shape[0]; Np_ = eng.padded_dim(n_)
                    th12 = list(np.linspace(-0.5, 0.5, 12))
                    FP12 = eng.ess_propose(F_, nu_[0], th12)
                    yy = imp._node_y(1, 0)
                    for md in (2, 2, 2, 0, 2, 1, 2):
                        eng.set_potrf_mode(md)
                        for label, A_, w_ in (('plan buffers', plan.A, plan.work), ('fresh buffers', eng.empty(12 * Np_ * Np_), eng.empty(int(plan.work.numel() // 8) + 16))):
                            A3 = A_.view(torch.float64)[:12 * Np_ * Np_].view(12, Np_, Np_) if A_.dtype != torch.float64 else A_[:12 * Np_ * Np_].view(12, Np_, Np_)
                            eng.kmatrix(nd2.name, FP12, np.asarray(nd2.input_dim, dtype=np.int32), imp._glob[(1, 0)], nd2.length, nd2.nugget[0], out=A3, full=False, Y=yy, batch=12)
                            ld_, info_ = eng.potrf(n_, A3, batch=12, work=w_)
                            print('   mode %d, %s: info %s' % (md, label, info_.cpu().numpy().tolist()))
                    eng.set_potrf_mode(mode)
                except Exception as ex4:
                    print('   diagnostics failed', repr(ex4))
                try:
                    imp.sample(burnin=10)
                    print('  retry of sample() passed', imp.stats)
                except Exception as ex2:
                    print('  retry failed too', repr(ex2), imp.stats)
                sys.exit(0)
            model._m_step()
        ti = tm = 0.0
        N = 8
        for _ in range(N):
            torch.cuda.synchronize(); t = time.perf_counter(); imp.sample(burnin=10); torch.cuda.synchronize(); ti += time.perf_counter() - t
            t = time.perf_counter(); model._m_step(); torch.cuda.synchronize(); tm += time.perf_counter() - t
        print('potrf mode %d, queued ESS %-5s: I-step %.1f ms, M-step %.1f ms, total %.1f ms' % (mode, queued, 1e3 * ti / N, 1e3 * tm / N, 1e3 * (ti + tm) / N), imp.stats, float(model.all_layer[1][0].length[0]))
