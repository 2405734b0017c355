#!/usr/bin/env python3
"""Timeline of ESS batches from a rocprofv3 kernel trace: per batch (ess_propose .. loglik_finish) kernel time + gaps,
and the idle time between consecutive batches."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-28:]) for r in csv.DictReader(open(f))))
batches, cur = [], None
for s, e, n in ev:
    if 'ess_propose' in n:
        cur = [(s, e, n)]
    elif cur is not None:
        cur.append((s, e, n))
        if 'loglik_finish' in n:
            batches.append(cur); cur = None
batches = batches[len(batches) // 3:]
span = sum(b[-1][1] - b[0][0] for b in batches) / len(batches)
busy = sum(sum(e - s for s, e, _ in b) for b in batches) / len(batches)
between = [batches[i + 1][0][0] - batches[i][-1][1] for i in range(len(batches) - 1)]
between = [g for g in between if g < 2e6]
agg = collections.defaultdict(lambda: [0.0, 0])
for b in batches:
    for s, e, n in b:
        agg[n][0] += e - s; agg[n][1] += 1
print('%d batches: span propose..finish %.1f us (kernels %.1f us), idle until the next batch %.1f us (median %.1f)' % (
    len(batches), span / 1e3, busy / 1e3, sum(between) / len(between) / 1e3, sorted(between)[len(between) // 2] / 1e3))
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
    print('%-30s %7.1f us per batch (%4.1f launches x %6.1f us)' % (n, t / len(batches) / 1e3, c / len(batches), t / c / 1e3))
