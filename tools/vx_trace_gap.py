"""Voxeliser launches in a rocprofv3 --kernel-trace run: per-call first-kernel / gap / last-kernel / span averages.
usage: python tools/vx_trace_gap.py DIR [skip_first_calls] [alg_bytes]
Pairs every vxl_keybin dispatch with the vxl_emit dispatch that follows it on the same queue.  Prints one JSON object with the
averages over all calls after `skip_first_calls`, over consecutive chunks (so that plain and event-carrying launches of bench.py
can be told apart by position), and — with alg_bytes — the roofline fraction of 8 TB/s the span corresponds to."""
import csv
import glob
import json
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
alg = float(sys.argv[3]) if len(sys.argv) > 3 else None
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "vxl_keybin" in n or "vxl_emit" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "k" if "vxl_keybin" in n else "e"))
rows.sort()
calls = []
i = 0
while i + 1 < len(rows):
    if rows[i][2] == "k" and rows[i + 1][2] == "e":
        k, e = rows[i], rows[i + 1]
        calls.append(((k[1] - k[0]) / 1e3, (e[0] - k[1]) / 1e3, (e[1] - e[0]) / 1e3, (e[1] - k[0]) / 1e3))
        i += 2
    else:
        i += 1
calls = calls[skip:]


def avg(cs):
    n = max(len(cs), 1)
    o = {"calls": len(cs), "keybin_us": sum(c[0] for c in cs) / n, "gap_us": sum(c[1] for c in cs) / n,
         "emit_us": sum(c[2] for c in cs) / n, "span_us": sum(c[3] for c in cs) / n}
    o["kernel_sum_us"] = o["keybin_us"] + o["emit_us"]
    if alg:
        o["frac_of_8TBs"] = alg / (o["span_us"] * 1e-6) / 8e12
    return o


out = {"all": avg(calls), "chunks_of_50": [avg(calls[j:j + 50]) for j in range(0, len(calls), 50)]}
print(json.dumps(out, indent=1))
