#!/usr/bin/env python3
"""One SI iteration of the bench model from a rocprofv3 --kernel-trace: the device's own account of the step, to be read beside bench.py's
step_split (host clock) -- iterations, I-step and M-step spans, kernel time by kernel, idle time by place, M-step rounds.

usage: analyze_step.py <rocprof output dir | trace.tsv.gz>  [skip_percent]
An iteration is cut at the first kernel of its I-step (the prior draws' trmv / the first ess_* kernel behind an M-step round); its I-step ends with
ess_end_kernel (+ mail_publish_kernel), its M-step is the run of rounds kmatrix_multi_kernel .. grad_final_multi_kernel that follows."""
import collections, csv, glob, gzip, sys

src = sys.argv[1]
if src.endswith('.gz'):
    ev = [l.rstrip('\n').split('\t') for l in gzip.open(src, 'rt')]
    ev = sorted((int(a), int(b), c) for a, b, c in ev)
else:
    f = glob.glob(src + '/**/*_kernel_trace.csv', recursive=True)[0]
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-48:]) for r in csv.DictReader(open(f)))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
ev = ev[int(len(ev) * skip / 100):]


def short(n):
    for k in ('potrf_mega_kernel', 'potrf_worker_kernel', 'potrf_gate_kernel', 'kmatrix_multi_kernel', 'kmatrix_kernel', 'grad_reduce_multi_kernel', 'grad_final_multi_kernel',
              'trmv_lower_kernel', 'copyBuffer', 'fillBuffer', 'direct_copy_kernel', 'mail_publish_kernel'):
        if k in n:
            return k
    return 'ess_*' if 'ess_' in n else n[-28:]


# ---- cut into iterations: an M-step is a maximal run of rounds (kmatrix_multi_kernel .. grad_final_multi_kernel) with no ess_* kernel inside; the I-step of an
#      iteration is everything between the previous M-step's last kernel and this M-step's first round, ending with the LAST ess_end_kernel before it (an I-step whose
#      device queue ran out of speculative batches is finished by the host and queued anew: several ess_end_kernel per I-step)
mblocks, cur = [], None
for j, e in enumerate(ev):
    n = e[2]
    if 'kmatrix_multi' in n and cur is None:
        cur = [j, j]
    if cur is not None:
        if 'ess_' in n or 'trmv_lower' in n:
            mblocks.append(cur); cur = None
        elif 'grad_final_multi' in n:
            cur[1] = j
if cur is not None:
    mblocks.append(cur)
iters = []
for (a0, a1), (b0, b1) in zip(mblocks, mblocks[1:]):
    between = ev[a1 + 1:b0]                       # the I-step of the iteration whose M-step is (b0, b1), with both boundaries
    ends_ = [j for j, e in enumerate(between) if 'ess_end_kernel' in e[2]]
    if not ends_:
        continue
    last_end = ends_[-1]
    while last_end + 1 < len(between) and 'mail_publish' in between[last_end + 1][2]:
        last_end += 1
    iters.append(dict(t_prev_end=between[last_end][1], pre=between[last_end + 1:], m=ev[b0:b1 + 1], i=between[:last_end + 1], t_m_prev_end=ev[a1][1],
                      queues=len(ends_)))
if len(iters) < 3:
    sys.exit('fewer than three iterations in the window')
print('%d iterations in the window (one ess_end_kernel each)' % len(iters))
acc = collections.defaultdict(float)
kern_i, kern_m = collections.defaultdict(lambda: [0.0, 0]), collections.defaultdict(lambda: [0.0, 0])
rounds, round_span, turn = 0, 0.0, []
for it in iters:
    m, i, pre = it['m'], it['i'], it['pre']
    acc['boundary M->I (last round of the previous M-step .. first I-step kernel)'] += (i[0][0] - it['t_m_prev_end']) if i else 0.0
    acc['I-step span (first kernel .. last ess_end_kernel)'] += (i[-1][1] - i[0][0]) if i else 0.0
    acc['boundary I->M (ess_end .. first M-step round: detach, diagnostics, set-up)'] += m[0][0] - it['t_prev_end']
    acc['M-step span (first round .. last round)'] += m[-1][1] - m[0][0]
    acc['I-steps whose device queue was re-queued (count per iteration)'] += 1e6 * (it['queues'] > 1)
    for s, e, n in m:
        k = kern_m[short(n)]; k[0] += e - s; k[1] += 1
    for s, e, n in i + pre:
        k = kern_i[short(n)]; k[0] += e - s; k[1] += 1
    # idle inside the two spans
    for name, seg in (('idle inside the M-step span (turn-arounds between rounds)', m), ('idle inside the I-step span', i)):
        end = seg[0][1] if seg else 0
        for s, e, n in seg[1:]:
            if s > end:
                acc[name] += s - end
            end = max(end, e)
    cur = None
    for s, e, n in m:
        if 'kmatrix_multi' in n:
            if cur is not None and False:
                pass
            cur = s
            if rounds and last_end is not None and s - last_end < 1e6:
                turn.append(s - last_end)
        if 'grad_final_multi' in n:
            rounds += 1
            round_span += e - cur
            last_end = e
    last_end = None
N = float(len(iters))
tot = sum(v for k, v in acc.items() if not k.startswith('idle') and not k.startswith('I-steps whose'))
print('per iteration (ms): sum of the four consecutive pieces = %.3f' % (tot / N / 1e6))
for k, v in acc.items():
    print('  %8.3f  %s' % (v / N / 1e6, k))
for title, kk in (('I-step kernels (incl. the boundary\'s copies)', kern_i), ('M-step kernels', kern_m)):
    print(title + ': ms per iteration, launches per iteration, us per launch')
    for n, (t, c) in sorted(kk.items(), key=lambda kv: -kv[1][0]):
        print('  %8.3f  %7.1f  %8.1f  %s' % (t / N / 1e6, c / N, t / c / 1e3, n))
print('M-step rounds per iteration %.1f, mean round span (kmatrix_multi .. grad_final_multi) %.1f us' % (rounds / N, round_span / max(rounds, 1) / 1e3))
if turn:
    turn.sort()
    print('turn-around between rounds (grad_final_multi end -> next kmatrix_multi start): median %.1f us, mean %.1f us, p90 %.1f us'
          % (turn[len(turn) // 2] / 1e3, sum(turn) / len(turn) / 1e3, turn[int(len(turn) * 0.9)] / 1e3))
