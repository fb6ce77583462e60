"""SECOND sparse backbone, training step shape: forward + backward (dgrad + wgrad) per 16 frames (ring clouds)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone, vfe
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.voxelizer import BatchVoxelizer
dev = torch.device("cuda:0"); B = 16
frames = [synth.cloud_ring(2000 + f) for f in range(B)]
o = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000).voxelize_frames(frames, device=dev)
bd = {"voxels": o["voxels"], "voxel_num_points": o["voxel_num_points"], "voxel_coords": o["voxel_coords"], "batch_size": B}
bd = vfe.MeanVFE(AttrDict(), 4)(bd)
m = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, [1408, 1600, 40]).to(dev).train()


def step():
    out = m(dict(bd))["encoded_spconv_tensor"].features
    loss = out.square().mean()
    m.zero_grad(set_to_none=True)
    loss.backward()


for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print(f"VoxelBackBone8x train-mode forward+backward: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per 16 frames")
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m(dict(bd))
    torch.cuda.synchronize()
    print(f"  forward only (train-mode BN, module sequence): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
