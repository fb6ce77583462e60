"""Prints a rocprofv3 *_kernel_stats.csv as a compact table.  usage: python tools/kstats.py DIR [filter]"""
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(d + "/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Name"]:
            print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} "
                  f"min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
