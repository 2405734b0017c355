#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into the text summary kept under profiles/."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('# rocprofv3 --kernel-trace --stats  (%s)' % ' '.join(sys.argv[2:]))
print('# %-72s %8s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', '%'))
for r in rows[:25]:
    print('%-74s %8s %12.3f %10.2f %10.2f %10.2f %6.1f' % (r['Name'][:74], r['Calls'], float(r['TotalDurationNs']) / 1e6,
          float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
print('# total kernel time %.3f ms over %d kernels' % (tot / 1e6, len(rows)))
