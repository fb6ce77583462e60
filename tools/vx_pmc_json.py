"""Builds profiles/rNN/voxelize_pmc.json from the four rocprofv3 PMC passes of tools/vx_pmc_collect.sh.
usage: python tools/vx_pmc_json.py gpurun_out"""
import collections, csv, glob, hashlib, json, os, sys

root = sys.argv[1]
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d, counter, skip=5):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, d) + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "vxl_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: sum(v for _, v in sorted(rows)[skip:]) / max(len(rows) - skip, 1) for k, rows in acc.items()}


out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv -- python3 tools/vx_bench.py --algos 3 --iters 10 [--resident] "
                  "(four separate passes, tools/vx_pmc_collect.sh); per-dispatch averages after the 5 warm-up calls",
       "workload": "16 frames x 20000 uniform points, PointPillar-KITTI grid, 256000 rows",
       "algorithmic_bytes_per_launch": 16 * 320000 + 256000 * 532,
       "fetch_correction": "MI355X_MICROARCH.md HBM: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads; the voxeliser's reads "
                           "are a mix of 16-B, 8-B and 4-B accesses (uncalibrated), so the read side is bracketed: raw .. 2 x raw",
       "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB"}
with open(os.path.join(here, "lidardetection_amd", "csrc", "voxelize.hip"), "rb") as fh:
    out["kernel_source_sha256_16"] = hashlib.sha256(fh.read()).hexdigest()[:16]
for mode in ("full", "resident"):
    f = per_kernel(f"vx_pmc_{mode}_fetch", "FETCH_SIZE")
    w = per_kernel(f"vx_pmc_{mode}_write", "WRITE_SIZE")
    fb, wb = sum(f.values()) * 1024, sum(w.values()) * 1024
    out[mode] = {"per_kernel_FETCH_SIZE_KiB": f, "per_kernel_WRITE_SIZE_KiB": w, "write_bytes_per_launch": wb,
                 "fetch_bytes_per_launch_raw": fb, "traffic_bytes_per_launch_low": wb + fb, "traffic_bytes_per_launch_high": wb + 2 * fb}
# bench.py reads the resident figures (the path its timed step runs)
out["traffic_bytes_per_launch_high"] = out["resident"]["traffic_bytes_per_launch_high"]
out["traffic_bytes_per_launch_low"] = out["resident"]["traffic_bytes_per_launch_low"]
print(json.dumps(out, indent=1))
