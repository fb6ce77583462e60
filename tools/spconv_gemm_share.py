"""How much of a sparse backbone's forward is implicit-GEMM time?  Times every ops.indice_conv_fused call of one forward with HIP
events (serialised), for the SECOND-KITTI (bs 16) and the NuScenes multi-head (bs 4) backbones."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth, pillar_ops
from lidardetection_amd.spconv import ops
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.voxelizer import BatchVoxelizer
dev = torch.device("cuda:0")
orig = ops.indice_conv_fused
log = []


def timed(feats, nbr, w, b, residual=None, relu=False, order=None, packed=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    out = orig(feats, nbr, w, b, residual, relu, order, packed)
    e1.record(); torch.cuda.synchronize()
    K, cin, cout = w.shape
    log.append((e0.elapsed_time(e1) * 1e3, 2.0 * float((nbr >= 0).sum()) * cin * cout, nbr.shape[0], cin, cout, K))
    return out


def run(name, cls, frames, vs, rng, P, maxv, C, grid):
    B = len(frames)
    o = BatchVoxelizer(vs, rng, P, maxv, C).voxelize_frames(frames, device=dev)
    feats = pillar_ops.mean_vfe(o["voxels"], o["voxel_num_points"])
    m = cls(AttrDict(), C, grid).to(dev).eval()
    bd = {"voxel_features": feats, "voxel_coords": o["voxel_coords"], "batch_size": B}
    with torch.no_grad():
        for _ in range(3): m(dict(bd))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): m(dict(bd))
        e1.record(); torch.cuda.synchronize()
        fwd = e0.elapsed_time(e1) / 5
        ops.indice_conv_fused = timed
        import lidardetection_amd.spconv.conv as cv
        log.clear()
        m(dict(bd))
        ops.indice_conv_fused = orig
    us = sum(l[0] for l in log); fl = sum(l[1] for l in log)
    print(f"{name}: {feats.shape[0]} voxels, forward {fwd:.2f} ms; {len(log)} implicit GEMMs {us / 1e3:.2f} ms alone = {fl / us / 1e6:.1f} TFLOP/s useful "
          f"({fl / us / 1e6 / 157.3 * 100:.1f} % of fp32 MFMA peak), {fl / 1e9:.1f} GFLOP")
    for l in log:
        print(f"     rows {l[2]:7d} {l[3]:3d}->{l[4]:3d} K {l[5]:2d}: {l[0]:7.1f} us {l[1] / l[0] / 1e6:6.1f} TF")


run("SECOND-KITTI VoxelBackBone8x bs16", spconv_backbone.VoxelBackBone8x, [synth.cloud_ring(2000 + f) for f in range(16)], synth.SEC_VOXEL,
    synth.SEC_RANGE, 5, 16000, 4, [1408, 1600, 40])
run("NuScenes VoxelResBackBone8x bs4", spconv_backbone.VoxelResBackBone8x, [synth.cloud_nus(4000 + f) for f in range(4)], synth.NUS_VOXEL,
    synth.NUS_RANGE, 10, 60000, 5, [1024, 1024, 40])
