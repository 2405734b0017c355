#!/usr/bin/env python3
"""Top rows of a rocprofv3 --stats --output-format csv kernel_stats file under the directory given."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print("%-72s calls %6s total %9.2f ms avg %8.1f us  %5s%%" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                               float(r["AverageNs"]) / 1e3, r["Percentage"]))
