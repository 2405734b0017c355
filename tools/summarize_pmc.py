#!/usr/bin/env python3
"""Per-kernel averages of one PMC counter from a rocprofv3 --pmc run (counter_collection.csv)."""
import csv, glob, sys, collections
d, counter = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
tot = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != counter:
        continue
    t = tot[r['Kernel_Name'].split('(')[0][-44:]]
    t[0] += float(r['Counter_Value'])
    t[1] += 1
for k, (v, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print('%-46s %8d launches  %14.1f per launch  %16.1f total' % (k, n, v / n, v))
