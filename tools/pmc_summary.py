"""Per-kernel averages of a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run.
usage: python tools/pmc_summary.py DIR [name-filter] [skip-first-N-dispatches-per-kernel]
Prints one JSON object: {kernel: {"dispatches": n, "avg_us": t, counter: average per dispatch, ...}}."""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        acc[short][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                              int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = {}
for k, cs in acc.items():
    o = {}
    for c, rows in cs.items():
        rows = sorted(rows)[skip:]
        if not rows:
            continue
        o["dispatches"] = len(rows)
        o["avg_us"] = sum(x[2] for x in rows) / len(rows) / 1e3
        o[c] = sum(x[1] for x in rows) / len(rows)
    out[k] = o
print(json.dumps(out, indent=1))
