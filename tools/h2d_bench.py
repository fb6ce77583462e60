"""PointPillar-KITTI step with the raw points starting in pinned HOST memory (PCIe-inclusive rate for DESIGN.md §4):
each step copies the 16 x 20 000 x 16 B point buffer H2D on the compute stream, then runs the same forward + NMS."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.pointpillar import PointPillarKITTI

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
B = 16
frames = [synth.cloud_uniform(1000 + f) for f in range(B)]
sizes = [len(f) for f in frames]
host = torch.from_numpy(np.concatenate(frames, 0)).pin_memory()
offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
pts = torch.empty_like(host, device=dev)
m = PointPillarKITTI(batch_size=B, max_voxels=16000, n_max=max(sizes), device=dev).randomize_for_bench(0)
with torch.no_grad():
    for _ in range(5):
        pts.copy_(host, non_blocking=True); m(pts, offs)
    for label, h2d in (("resident", False), ("H2D each step", True)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            if h2d: pts.copy_(host, non_blocking=True)
            m(pts, offs)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"{label:14s}: {B * 20 / dt:8.1f} frames/s ({dt / 20 * 1e3:.3f} ms/step)")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): pts.copy_(host, non_blocking=True)
    e1.record(); torch.cuda.synchronize()
    print(f"H2D copy alone: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us for {host.numel() * 4 / 1e6:.2f} MB "
          f"({host.numel() * 4 / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9:.1f} GB/s)")
