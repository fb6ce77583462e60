"""SECOND sparse backbone: a few forwards, idle, one more (for tools/ktrace_last.py)."""
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
name = sys.argv[1] if len(sys.argv) > 1 else "VoxelBackBone8x"
m = getattr(spconv_backbone, name)(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
with torch.no_grad():
    for _ in range(3): m(dict(bd))
    torch.cuda.synchronize(); time.sleep(1.0)
    out = m(dict(bd))
    torch.cuda.synchronize()
print("voxels in", bd["voxel_features"].shape, "out", out["encoded_spconv_tensor"].features.shape)
