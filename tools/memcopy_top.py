#!/usr/bin/env python3
"""Sizes, directions and counts of the memory copies in a rocprofv3 --memory-copy-trace CSV under the directory given."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[int(len(rows) * skip / 100):]
agg = collections.defaultdict(lambda: [0, 0, 0])
for r in rows:
    size = int(r.get('Size', r.get('size', 0)) or 0)
    bucket = 1 << max(0, size - 1).bit_length() if size else 0
    k = (r.get('Direction', r.get('direction', '?')), bucket)
    a = agg[k]
    a[0] += 1
    a[1] += size
    a[2] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print(list(rows[0].keys()))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:16]:
    print('%-28s <= %10d B   n %6d   total %9.1f MB   time %8.2f ms' % (k[0], k[1], a[0], a[1] / 1e6, a[2] / 1e6))
