"""Voxeliser on the other BASELINE shapes: SECOND-KITTI (P=5, 0.05 m voxels, ring clouds) and NuScenes (30k pts, 5 features,
P=10, 60k voxels, bs 4) — time per launch and algorithmic GB/s (16N*... per SURVEY §8d with the actual C)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.voxelizer import BatchVoxelizer
dev = torch.device("cuda:0")


def run(name, frames, voxel, rng, P, maxv, C):
    sizes = [len(f) for f in frames]
    pts = torch.from_numpy(np.concatenate(frames)).to(dev)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    vz = BatchVoxelizer(voxel, rng, P, maxv, C)
    out = vz.alloc_outputs(len(frames), dev)
    for _ in range(5): vz(pts, offs, max(sizes), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): vz(pts, offs, max(sizes), out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    rows = int(out["voxel_offsets"][-1])
    alg = 4 * C * sum(sizes) + rows * (P * C * 4 + 20)
    print(f"{name}: {ms * 1e3:.1f} us/launch, {len(frames)} frames, rows {rows}, err flag {vz.error_flag(len(frames), max(sizes), dev)}, "
          f"{alg / ms / 1e6:.0f} GB/s algorithmic")


run("SECOND-KITTI ring bs16 (P=5)", [synth.cloud_ring(2000 + f) for f in range(16)], synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000, 4)
run("PointPillar ring bs16 (P=32)", [synth.cloud_ring(2000 + f) for f in range(16)], synth.PP_VOXEL, synth.PP_RANGE, 32, 16000, 4)
run("NuScenes uniform bs4 (N=30k, C=5, P=10, 60k voxels)", [synth.cloud_nus(4000 + f) for f in range(4)], synth.NUS_VOXEL, synth.NUS_RANGE, 10, 60000, 5)
