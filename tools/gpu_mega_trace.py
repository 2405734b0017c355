"""Phase stamps of the one-launch factorisation (dgpamd_debug_trace): the chain of matrix 0 per block step and the first
tasks of one worker.  usage: gpu_mega_trace.py n B [inv]"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine
from dgp_amd._lib import lib

eng = Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
inv = len(sys.argv) > 3 and sys.argv[3] == 'inv'
Np = eng.padded_dim(n)
rng = np.random.default_rng(0)
X = eng.tensor(rng.uniform(size=(B, n, 5)))
G = eng.tensor(rng.uniform(size=(n, 5)))
y = eng.tensor(rng.normal(size=n))
A = eng.empty(B, Np, Np)
work = eng.potrf_workspace(n, B)
T, S = eng.empty(B, Np, Np), eng.empty(B, Np, Np)
tr = torch.zeros(8192, dtype=torch.int64, device=A.device)
eng.set_potrf_mode(1)
for rep in range(3):
    eng.kmatrix('matern2.5', X, None, G, [1.0], 1e-6, out=A, full=False, Y=y, batch=B)
    if rep == 2:
        lib.dgpamd_debug_trace(eng.h, C.c_void_p(tr.data_ptr()))
    if inv:
        eng.potrf_inv(n, A, T, S, batch=B, work=work)
    else:
        eng.potrf(n, A, batch=B, work=work)
torch.cuda.synchronize()
lib.dgpamd_debug_trace(eng.h, None)
raw = tr.cpu().numpy().astype(np.float64) / 100.0   # us
nbk = Np // 64
t = raw[:16 * nbk].reshape(-1, 16)
print('n=%d B=%d inv=%s: chain of matrix 0, us' % (n, B, inv))
print(' k | factor  publish W  wait inputs  solve  update | step')
for k in range(nbk - 1):
    r = t[k]
    print('%2d | %5.1f  %5.1f  %5.1f  %5.1f  %5.1f | %5.1f' % (k, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], t[k + 1][0] - r[0]))
print('total chain %.1f us' % (t[nbk - 1][1] - t[0][0]))
if t[5][6] > 0:   # the second chain form's finer stamps (round 4)
    print('block 5 after its factorisation (us): W drained + panel tile landed + staged %.2f | next diagonal tile requested, factor tile stored %.2f | solve MFMAs %.2f | barrier %.2f | exchange, panel stores, barrier %.2f | tile landed + transposed + update MFMAs (wave 0) %.2f' % (
        t[5][2] - t[5][1], t[5][6] - t[5][2], t[5][4] - t[5][6], t[5][7] - t[5][4], t[5][8] - t[5][7], t[5][5] - t[5][8]))
d = raw[1024:1024 + 16 * nbk].reshape(-1, 16)
r = d[5]
print('block 5, diagonal factor (us from its start): ' + ' | '.join(
    '16-block %d: start %.2f, factored %.2f' % (J, r[2 * J] - r[0], r[2 * J + 1] - r[0]) for J in range(4)) + ' | end %.2f' % (r[8] - r[0]))
# the fused look-ahead task of block k (T_LOOK, round 4), relative to the END of the chain's factorisation of block k
lk = raw[7000:7000 + 8 * nbk].reshape(-1, 8)
if lk[:, 0].any():
    print('look-ahead tasks of block k, us after the chain factored block k: pulled | own inputs | W_k seen | S solved | P, Q flags seen | Q published | D published | S published || chain: W_k published, next factor done')
    for k in range(2, min(nbk - 2, 12)):
        if lk[k][0] == 0:
            continue
        t0 = t[k][1]
        print('%2d | %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f || %5.1f %5.1f' % ((k,) + tuple(lk[k][i] - t0 for i in range(8)) + (t[k][2] - t0, t[k + 1][1] - t0)))
w = raw[2048:2048 + 640].reshape(-1, 8)
iw = tr.cpu().numpy()[2048:2048 + 640].reshape(-1, 8)
print('one worker, per task (us): kind panels | wait inputs  compute  store  publish | total   gap to next pull')
kinds = {0: 'store', 1: 'solve', 3: 'tdiag'}
for i in range(len(w)):
    if w[i][0] == 0:
        break
    code = int(iw[i][5])
    gap = w[i + 1][0] - w[i][4] if i + 1 < len(w) and w[i + 1][0] > 0 else float('nan')
    print('%3d %5s %d | %5.1f  %5.1f  %5.1f  %5.1f | %5.1f   %5.1f' % (i, kinds.get(code & 15, '?'), (code >> 8) & 255, w[i][1] - w[i][0], w[i][2] - w[i][1],
                                                              w[i][3] - w[i][2], w[i][4] - w[i][3], w[i][4] - w[i][0], gap))
tot = {'wait': 0.0, 'compute': 0.0, 'store': 0.0, 'publish': 0.0, 'gap': 0.0}
nt = 0
for i in range(len(w)):
    if w[i][0] == 0:
        break
    nt += 1
    if min(w[i][1], w[i][2], w[i][3]) > 0:   # (a T[k][k] = W_k^T task only stamps its two ends: all of it counts as store)
        tot['wait'] += w[i][1] - w[i][0]; tot['compute'] += w[i][2] - w[i][1]; tot['store'] += w[i][3] - w[i][2]; tot['publish'] += w[i][4] - w[i][3]
    else:
        tot['store'] += w[i][4] - w[i][0]
    if i + 1 < len(w) and w[i + 1][0] > 0:
        tot['gap'] += w[i + 1][0] - w[i][4]
span = w[nt - 1][4] - w[0][0] if nt else 0.0
print('this worker: %d tasks over %.1f us: ' % (nt, span) + ', '.join('%s %.1f%%' % (k, 100 * v / span) for k, v in tot.items()))
print('first task pulled %.1f us after the chain started, last published %.1f us after' % (w[0][0] - t[0][0], w[nt - 1][4] - t[0][0]))

# pool-wide time split per 100 us of the launch (every worker's tasks, bucketed by the task's start)
aw, ac, ar = raw[6000:6032], raw[6032:6064], raw[6064:6096]
print('all workers, us per 100-us bucket of task start time: waiting for inputs | arithmetic | store / W wait / publish')
for i in range(32):
    tot = aw[i] + ac[i] + ar[i]
    if tot > 0:
        print('%4d-%4d us: %9.0f %9.0f %9.0f   (%4.1f%% / %4.1f%% / %4.1f%%)' % (100 * i, 100 * i + 100, aw[i], ac[i], ar[i], 100 * aw[i] / tot, 100 * ac[i] / tot, 100 * ar[i] / tot))
T = aw.sum() + ac.sum() + ar.sum()
if T > 0:
    print('total worker time %.0f us: waiting %.1f%%, arithmetic %.1f%%, rest %.1f%%' % (T, 100 * aw.sum() / T, 100 * ac.sum() / T, 100 * ar.sum() / T))
names = ['A update', 'A solve', 'T update', 'T solve', 'K^-1 update', '-']
print('by kind of task: count | mean waiting, arithmetic, rest (us) | share of all worker time')
for i, nm in enumerate(names):
    w_, c_, r_, n_ = raw[6100 + 4 * i:6104 + 4 * i]
    if n_ > 0:
        print('%-12s %6d | %6.1f %6.1f %6.1f | %4.1f%%' % (nm, n_, w_ / n_, c_ / n_, r_ / n_, 100 * (w_ + c_ + r_) / T))
