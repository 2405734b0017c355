#!/usr/bin/env python3
"""Kernel sequences around the long idle gaps of a rocprofv3 --kernel-trace CSV (second half of the trace).
usage: analyze_context.py <dir> [min gap us]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('<')[0][-30:]) for r in csv.DictReader(open(f))))
ev = ev[len(ev) // 2:]
ming = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 150e3
seqs = collections.defaultdict(lambda: [0, 0.0])
for i in range(4, len(ev) - 4):
    g = ev[i + 1][0] - ev[i][1]
    if g > ming:
        key = ' | '.join(e[2] for e in ev[i - 3:i + 1]) + '  ==>  ' + ' | '.join(e[2] for e in ev[i + 1:i + 5])
        seqs[key][0] += 1
        seqs[key][1] += g
for s, (c, t) in sorted(seqs.items(), key=lambda kv: -kv[1][1])[:12]:
    print('%3d x %7.1f us   %s' % (c, t / c / 1e3, s))
