#!/usr/bin/env python3
"""All PMC counters of ONE kernel (name substring) from rocprofv3 --pmc runs: per-launch averages.
usage: pmc_kernel.py <kernel substring> <dir> [<dir> ...]"""
import csv, glob, sys, collections
name = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        tot = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if name in r['Kernel_Name']:
                t = tot[r['Counter_Name']]
                t[0] += float(r['Counter_Value']); t[1] += 1
        for k, (v, n) in sorted(tot.items()):
            print('%-32s %6d launches  %18.1f per launch' % (k, n, v / n))
