#!/usr/bin/env python3
"""Mean duration of every kernel of a repeating launch sequence and the mean gap in front of it (rocprofv3 kernel trace;
last two thirds of the trace).  usage: analyze_trace_seq.py <dir>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-34:]) for r in csv.DictReader(open(f))))
ev = ev[len(ev) // 3:]
agg = collections.OrderedDict()
for (s0, e0, n0), (s, e, n) in zip(ev, ev[1:]):
    a = agg.setdefault((n0, n), [0.0, 0.0, 0])
    a[0] += e - s; a[1] += s - e0; a[2] += 1
for (n0, n), (d, g, c) in agg.items():
    print('%-36s -> %-36s x%4d: gap %7.1f us, then %7.1f us' % (n0, n, c, g / c / 1e3, d / c / 1e3))
print('span per repetition: %.1f us' % ((ev[-1][1] - ev[0][0]) / 1e3 / max(1, sum(1 for _, _, n in ev if 'potrf_mega' in n))))
