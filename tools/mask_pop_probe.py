"""How the mask-ordered row tiles of the SECOND stack are distributed in cost (offsets per 128-row workgroup) along the launch order."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth, spconv
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone, vfe
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.voxelizer import BatchVoxelizer
from lidardetection_amd.spconv import ops
dev = torch.device("cuda:0"); B = 16
frames = [synth.cloud_ring(2000 + f) for f in range(B)]
o = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000).voxelize_frames(frames, device=dev)
bd = {"voxels": o["voxels"], "voxel_num_points": o["voxel_num_points"], "voxel_coords": o["voxel_coords"], "batch_size": B}
bd = vfe.MeanVFE(AttrDict(), 4)(bd)
m = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
with torch.no_grad():
    out = m(dict(bd))
d = out["encoded_spconv_tensor"].indice_dict
for key in ("subm2", "subm3", "subm4"):
    nbr = d[key]["nbr"]
    masks, order = ops.mask_order(nbr)
    mo = masks[order.long()].cpu().numpy().astype(np.uint32)
    n = len(mo) // 128 * 128
    wg = np.bitwise_or.reduce(mo[:n].reshape(-1, 128), axis=1)
    pc = np.array([bin(int(v)).count("1") for v in wg])
    q = len(pc) // 8
    print(key, "workgroups", len(pc), "offsets per workgroup: mean", pc.mean().round(2), "by eighth of the launch order:", [float(pc[i*q:(i+1)*q].mean().round(1)) for i in range(8)])
