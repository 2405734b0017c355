"""Step log of the Matern pair kernel (linkgp_Jsep_kernel<2, true>, DGPAMD_JSEP_LOG=1): for every workgroup of ONE launch its start / end
(100-MHz clock and shader cycles), the CU it ran on, and for its first 48 steps every wave's arrival at and departure from the step's
barrier.  Written to an .npz; the summary printed here: occupancy of the workgroup slots over the launch (tail), time per step,
share of a wave's time spent waiting at the barrier, by order class.   usage: gpu_pair_steplog.py out.npz [n Dw M]"""
import os, sys, ctypes as C
os.environ['DGPAMD_JSEP_LOG'] = '1'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgp_amd.ops import Engine, cell_order
from dgp_amd._lib import lib

out = sys.argv[1]
n, Dw, M = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (2000, 5, 256)
eng = Engine(0)
rng = np.random.default_rng(5)
W = rng.normal(size=(n, Dw))
W = W[cell_order(W)]
G = rng.normal(size=(n, 8)) / np.sqrt(n)
Rinv, ry = G @ G.T + np.eye(n), rng.normal(size=n)
m, v = rng.normal(size=(M, Dw)), rng.uniform(0.01, 0.4, size=(M, Dw))
dm, dv, dW, dR, dry = eng.tensor(m), eng.tensor(v), eng.tensor(W), eng.tensor(Rinv), eng.tensor(ry)
length = np.array([2.5])
run = lambda: eng.linkgp_predict('matern2.5', dm, dv, None, dW, None, length, dR, n, dry, 1.3, 1e-4)
run(); torch.cuda.synchronize()
nb = (n + 63) // 64
ntiles, tb, LW = nb * (nb + 1) // 2, (M + 31) // 32, 8 + 48 * 4 * 2
words = 64 + ntiles * tb * LW
log = torch.zeros(words, dtype=torch.int64, device=dW.device)
lib.dgpamd_debug_tasklog(eng.h, C.c_void_p(log.data_ptr()), words)
run(); torch.cuda.synchronize()
lib.dgpamd_debug_tasklog(eng.h, None, 0)
L = log.cpu().numpy()[64:].reshape(ntiles * tb, LW)
np.savez_compressed(out, log=L, n=n, Dw=Dw, M=M, ntiles=ntiles, tb=tb)

hd = L[:, :8]
t0, t1 = hd[:, 0], hd[:, 1]
span = (t1.max() - t0.min()) / 100.0   # us
busy = (t1 - t0).sum() / 100.0
cyc = (hd[:, 3] - hd[:, 2]) / ((t1 - t0) / 100.0)   # shader cycles per us
hw, xcc = hd[:, 4], hd[:, 5]
cu = (xcc & 0xf) * 1024 + ((hw >> 13) & 7) * 64 + ((hw >> 8) & 15) * 4 + ((hw >> 12) & 1)   # xcc, se, cu (, sh)
ncu = len(np.unique(cu))
print('n=%d Dw=%d M=%d: %d workgroups on %d CUs, launch span %.0f us; sum of workgroup times / (span x 2 x CUs) = %.3f; clock %.0f MHz'
      % (n, Dw, M, len(L), ncu, span, busy / (span * 2 * ncu), np.median(cyc)))
dur = (t1 - t0) / 100.0
setup = (t0 - hd[:, 7]) / 100.0
# per CU slot: the gap between one workgroup's end and the next one's entry
gaps = []
for c in np.unique(cu):
    m = cu == c
    ent, en = np.sort(hd[m, 7]), np.sort(t1[m])
    # two slots per CU: pair every entry (but the first two) with the earliest end not yet used
    ends = list(en)
    for e in ent[2:]:
        prev = [x for x in ends if x <= e]
        if prev:
            gaps.append((e - prev[0]) / 100.0)
            ends.remove(prev[0])
print('workgroup set-up (entry -> first step): median %.1f us, mean %.1f; slot idle between an end and the next entry on that CU: median %.1f us (mean %.1f, %d pairs)'
      % (np.median(setup), setup.mean(), np.median(gaps) if gaps else float('nan'), np.mean(gaps) if gaps else float('nan'), len(gaps)))
print('workgroup time: median %.0f us, 5%% %.0f, 95%% %.0f; last start at %.0f us of the span'
      % (np.median(dur), np.percentile(dur, 5), np.percentile(dur, 95), (t0.max() - t0.min()) / 100.0))
# slots busy over time (10 bins)
edges = np.linspace(t0.min(), t1.max(), 11)
occ = [(np.clip(np.minimum(t1, edges[i + 1]) - np.maximum(t0, edges[i]), 0, None)).sum() / ((edges[i + 1] - edges[i]) * 2 * ncu) for i in range(10)]
print('slot occupancy by tenth of the span:', ' '.join('%.2f' % o for o in occ))
S = L[:, 8:].reshape(len(L), 48, 4, 2)
arr, lv = S[..., 0], S[..., 1] & ((1 << 60) - 1)
cls = (S[..., 1] >> 60) & 3
ok = np.ones(len(L), dtype=bool)   # (every chunk of this run is full: 32 test points x Dw steps >= 48)
arr, lv, cls = arr[ok], lv[ok], cls[ok]
wait = lv - arr                       # cycles at the barrier (incl. the vmcnt wait in front of it)
comp = arr[:, 1:] - lv[:, :-1]        # barrier departure -> next arrival: the wave's step
stept = lv[:, 1:] - lv[:, :-1]
mid = slice(8, 47)
print('steps 8-46 of every workgroup: step %.0f cycles (median; mean %.0f), of which at the barrier %.0f (mean %.0f) = %.1f %%'
      % (np.median(stept[:, mid]), stept[:, mid].mean(), np.median(wait[:, mid]), wait[:, mid].mean(), 100.0 * wait[:, mid].mean() / stept[:, mid].mean()))
c = cls[:, :-1]
for k, name in ((0, 'mixed'), (1, 'class 1'), (2, 'class 2')):
    sel = c[:, mid] == k
    if sel.any():
        print('  %-8s %5.1f %% of wave-steps: compute %.0f cycles (median; mean %.0f), then waits %.0f (mean)'
              % (name, 100.0 * sel.mean(), np.median(comp[:, mid][sel]), comp[:, mid][sel].mean(), wait[:, 1:][:, mid][sel].mean()))
# the last wave to arrive waits ~0: how long do the others wait, and is the slowest wave the one with the most work?
last = arr[:, mid].argmax(-1)
print('last wave to arrive, share per wave: ' + ' '.join('%.2f' % (last == w).mean() for w in range(4)))
spread = arr[:, mid].max(-1) - arr[:, mid].min(-1)
print('arrival spread within a workgroup: median %.0f cycles, mean %.0f; departure spread %.0f'
      % (np.median(spread), spread.mean(), (lv[:, mid].max(-1) - lv[:, mid].min(-1)).mean()))
allmixed = (cls[:, mid] == 0).all(-1)
print('steps in which all four waves are mixed: %.1f %%: step %.0f cycles, arrival spread %.0f'
      % (100.0 * allmixed.mean(), stept[:, mid][allmixed[:, :stept[:, mid].shape[1]]].mean() if allmixed.any() else 0, spread[allmixed].mean() if allmixed.any() else 0))
