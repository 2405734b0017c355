#!/usr/bin/env python3
"""Timeline of M-step rounds from a rocprofv3 kernel trace: per round (kmatrix_multi .. grad_final_multi) the kernel
durations and the gaps between them."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-28:]) for r in csv.DictReader(open(f))))
rounds = []
cur = None
for s, e, n in ev:
    if 'kmatrix_multi' in n:
        cur = [(s, e, n)]
    elif cur is not None:
        cur.append((s, e, n))
        if 'grad_final_multi' in n:
            rounds.append(cur); cur = None
print(len(rounds), 'rounds')
agg = collections.defaultdict(lambda: [0.0, 0])
tot = gaps = 0.0
for r in rounds[len(rounds) // 3:]:
    tot += r[-1][1] - r[0][0]
    for i, (s, e, n) in enumerate(r):
        a = agg[n]; a[0] += e - s; a[1] += 1
        if i: gaps += max(0, s - r[i - 1][1])
nr = len(rounds) - len(rounds) // 3
print('mean span kmatrix..grad_final %.1f us, gaps inside %.1f us' % (tot / nr / 1e3, gaps / nr / 1e3))
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('%-30s %6.1f us per round (%4.1f launches x %6.1f us)' % (n, t / nr / 1e3, c / nr, t / c / 1e3))
# turn-around between consecutive rounds of one M-step: grad_final end -> next kmatrix_multi start (D2H copy, host, H2D copy)
ta = [b[0][0] - a[-1][1] for a, b in zip(rounds, rounds[1:]) if b[0][0] - a[-1][1] < 1e6]
if ta:
    ta.sort()
    print('turn-around between rounds: median %.1f us, mean %.1f us, p90 %.1f us (%d gaps)'
          % (ta[len(ta) // 2] / 1e3, sum(ta) / len(ta) / 1e3, ta[int(len(ta) * 0.9)] / 1e3, len(ta)))
