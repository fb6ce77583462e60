"""Micro-benchmark of the voxeliser alone (HIP events on the launch stream).
usage: python tools/vx_bench.py [--cloud uniform|ring] [--iters 100]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth  # noqa: E402
from lidardetection_amd.voxelizer import BatchVoxelizer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cloud", default="uniform")
ap.add_argument("--iters", type=int, default=100)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--algos", default="3,2")
ap.add_argument("--resident", action="store_true", help="resident output buffer (include/lidar_hip.h algo 4)")
ap.add_argument("--flush", action="store_true", help="stream 1 GiB through the caches before every call (the state the voxeliser "
                "finds inside a detector step: points, workspace and output buffer in HBM, not in L2 / Infinity Cache)")
a = ap.parse_args()
dev = torch.device("cuda:0")
gen = synth.cloud_uniform if a.cloud == "uniform" else synth.cloud_ring
frames = [gen((1000 if a.cloud == "uniform" else 2000) + f) for f in range(a.batch)]
sizes = [len(f) for f in frames]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
for algo in [int(x) for x in a.algos.split(",")]:
    vz = BatchVoxelizer(synth.PP_VOXEL, synth.PP_RANGE, 32, 16000, algo=algo)
    out = vz.alloc_outputs(a.batch, dev)
    for _ in range(5):
        vz(pts, offs, max(sizes), out=out, resident=a.resident)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        vz(pts, offs, max(sizes), out=out, resident=a.resident)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    if a.flush:
        junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
        for k, (x0, x1) in enumerate(evs):
            junk.fill_(float(k))
            x0.record()
            vz(pts, offs, max(sizes), out=out, resident=a.resident)
            x1.record()
        torch.cuda.synchronize()
        ms = float(np.median([x0.elapsed_time(x1) for x0, x1 in evs]))
        del junk
    rows = int(out["voxel_offsets"][-1])
    alg = 16 * sum(sizes) + rows * (32 * 4 * 4 + 20)
    print(f"algo {algo}{' resident' if a.resident else ''} cloud {a.cloud}{' (cold caches, event bracket per call)' if a.flush else ''}: {ms*1e3:.1f} us/launch, rows {rows}, {alg/ms/1e6:.0f} GB/s algorithmic "
          f"({alg/ms/1e6/8000*100:.1f}% of 8 TB/s)")
