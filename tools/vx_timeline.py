"""Prints the per-kernel timeline (start/end relative, us) of the last voxelize launch in a rocprofv3 kernel trace."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "vxl_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-5:]
t0 = min(int(r["Start_Timestamp"]) for r in last)
for r in last:
    print(f"{r['Kernel_Name'][:34]:34s} start {(int(r['Start_Timestamp'])-t0)/1e3:7.1f} end {(int(r['End_Timestamp'])-t0)/1e3:7.1f} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:6.1f}")
