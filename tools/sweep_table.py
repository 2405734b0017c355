"""Table of a tools/gpu_r5_sweep.sh output file: one row per setting, potrf / potrf_inv of the one-launch kernel (ms) per batch size."""
import re, sys
cur, rows = None, {}
for l in open(sys.argv[1]):
    if l.startswith("=="):
        cur = l[3:].strip()
        while cur in rows:
            cur += " (again)"
        rows[cur] = {}
    m = re.match(r"B=\s*(\d+)\s+mode0\s+\S+\s+\S+\s+\|\s+mode1\s+(\S+)\s+(\S+)", l)
    if m:
        rows[cur][int(m.group(1))] = (float(m.group(2)), float(m.group(3)))
bs = sorted({b for v in rows.values() for b in v})
w = max(len(k) for k in rows) + 2
print("%-*s" % (w, "setting") + "".join("  B=%-2d potrf   inv" % b for b in bs))
for k, v in rows.items():
    print("%-*s" % (w, k) + "".join("      %.3f %.3f" % v[b] if b in v else " " * 18 for b in bs))
