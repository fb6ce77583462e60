"""Per-kernel timeline from a rocprofv3 --kernel-trace run.
usage: python tools/ktrace_last.py DIR [max_name_chars] [MARKER]
Without MARKER: the LAST burst (everything after the last idle gap >= 0.5 s; the profiled script sleeps before its
final iteration).  With MARKER: one period = from the second-to-last kernel whose name contains MARKER to the last."""
import csv
import glob
import sys

d = sys.argv[1]
w = int(sys.argv[2]) if len(sys.argv) > 2 else 90
marker = sys.argv[3] if len(sys.argv) > 3 else None
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if marker:
    hits = [i for i, r in enumerate(rows) if marker in r[2]]
    last = rows[hits[-2]:hits[-1]]
else:
    gap_i, gap = 0, -1
    for i in range(1, len(rows)):
        g = rows[i][0] - rows[i - 1][1]
        if g > 0.5e9 or (gap < 0.5e9 and g > gap):   # the last idle gap of >= 0.5 s, else the largest one
            gap, gap_i = g, i
    last = rows[gap_i:]
t0 = last[0][0]
busy, prev_end = 0, t0
for s, e, n in last:
    busy += e - s
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  {n[:w]}")
    prev_end = e
print(f"kernels {len(last)}  busy {busy / 1e3:.1f} us  span {(last[-1][1] - t0) / 1e3:.1f} us")
