#!/usr/bin/env python3
"""GPU idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV, grouped by (previous -> next) kernel."""
import csv, glob, sys, collections
if sys.argv[1].endswith('.gz'):   # (start, end, name) rows as tools/gpu_round6_final.sh keeps them
    import gzip
    ev = sorted((int(a), int(b), c[-40:]) for a, b, c in (l.rstrip('\n').split('\t') for l in gzip.open(sys.argv[1], 'rt')))
else:
    f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]) for r in rows))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # ignore the first `skip` fraction (warm-up / construction)
ev = ev[int(len(ev) * skip / 100):]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
end = ev[0][1]
prev = ev[0][2]
for s, e, name in ev[1:]:
    if s > end:
        g = gaps[(prev, name)]
        g[0] += s - end
        g[1] += 1
    if e > end:
        end, prev = e, name
print('span %.1f ms, kernel busy (sum) %.1f ms, idle %.1f ms' % (span / 1e6, busy / 1e6, sum(g[0] for g in gaps.values()) / 1e6))
# SI iterations in the window: an I-step may be queued more than once (an update left open by a queue is continued by the next: several ess_end_kernel), so an
# iteration is counted where an M-step begins -- the first kmatrix_multi_kernel behind an ess_end_kernel (tools/analyze_step.py cuts the trace the same way)
nend, armed = 0, False
for _, _, nm in ev:
    if 'ess_end_kernel' in nm:
        armed = True
    elif armed and 'kmatrix_multi' in nm:
        nend += 1
        armed = False
if nend:   # the device's own work per iteration, whatever the profiler does to the host
    print('%d SI iterations in this window: %.1f ms of kernel time per iteration (the span per iteration under the profiler is %.1f ms; without it the iteration takes what the line above the table says)' % (nend, busy / 1e6 / nend, span / 1e6 / nend))
for (a, b), (t, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:18]:
    print('%8.2f ms  %6d x %7.1f us   %s -> %s' % (t / 1e6, n, t / n / 1e3, a, b))
if len(sys.argv) > 3:   # context of the gaps longer than argv[3] microseconds: the kernels either side, times relative to the gap's start
    thr = float(sys.argv[3]) * 1e3
    limit = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    shown = 0
    for i in range(1, len(ev)):
        if ev[i][0] - max(e for _, e, _ in ev[max(0, i - 8):i]) > thr and shown < limit:
            t0 = max(e for _, e, _ in ev[max(0, i - 8):i])
            print('--- gap of %.0f us' % ((ev[i][0] - t0) / 1e3))
            for s, e, nm in ev[max(0, i - 4):i + 10]:
                print('   %9.1f .. %9.1f us  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, nm))
            shown += 1
