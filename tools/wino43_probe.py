"""Timing ablations of csrc/wino43_conv.hip (-DWINO43_PROBE=bits builds: WRONG results by design) — which part of a chunk is exposed?
usage (GPU box): python tools/wino43_probe.py "0 1 2 4 8 16 31" [B,C,H,W ...]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "lidardetection_amd", "csrc")
F = "--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -Wno-inline-asm -mllvm -pragma-unroll-threshold=4000000".split()
objs = [os.path.join(C, o) for o in os.listdir(C) if o.endswith(".o") and o != "wino43_conv.o" and ".vxl_nowait." not in o]
names = {1: "no input DMA", 2: "no filter loads", 4: "no patch reads/transform", 8: "no epilogue", 16: "no barrier", 32: "no transform arithmetic + V writes", 128: "no patch reads", 256: "no global stores", 512: "no load drain before the stores"}
for tok in sys.argv[1].split():                 # "bits" or "bits:EXTRA_DEFINE=val" (e.g. 0:WINO_SCHED=0 for an A/B of two schedules on one box)
    v, extra = (tok.split(":", 1) + [""])[:2]
    v = int(v)
    so = f"/tmp/liblidar_wino43_probe_{tok.replace(':', '_').replace('=', '_').replace(',', '_')}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", *F, f"-DWINO43_PROBE={v}", *([f"-D{x}" for x in extra.split(",")] if extra else []), "-c", os.path.join(C, "wino43_conv.hip"),
                           "-o", f"/tmp/wino43_probe_{v}.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs, f"/tmp/wino43_probe_{v}.o"])
    r = subprocess.run([sys.executable, os.path.join(R, "tools", "wino_bench.py"), *sys.argv[2:]], env=dict(os.environ, LIDAR_HIP_SO=so, WINO_BENCH_ONLY="1"),
                       capture_output=True, text=True)
    what = " + ".join(n for b, n in names.items() if v & b) or "product kernel"
    print(f"== probe {tok}: {what}")
    for ln in r.stdout.splitlines():
        if "F(4x4)" in ln:
            print("   ", ln.split("|")[0])
    if r.returncode:
        print(r.stderr[-400:])
