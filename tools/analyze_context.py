#!/usr/bin/env python3
"""Kernel sequences around the idle gaps of one (previous -> next) kernel pair in a rocprofv3 --kernel-trace CSV.
usage: analyze_context.py <dir> <prev substring> <next substring> [min gap us]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-36:]) for r in csv.DictReader(open(f))))
a, b = sys.argv[2], sys.argv[3]
ming = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 20e3
seqs = collections.Counter()
for i in range(3, len(ev) - 3):
    if a in ev[i][2] and b in ev[i + 1][2] and ev[i + 1][0] - ev[i][1] > ming:
        seqs[' | '.join(e[2] for e in ev[i - 3:i + 1]) + '  ==gap==>  ' + ' | '.join(e[2] for e in ev[i + 1:i + 4])] += 1
for s, c in seqs.most_common(8):
    print(c, 'x', s)
