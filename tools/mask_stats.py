"""Distinct neighbour-offset masks per rulebook table of the SECOND-KITTI backbone (bs 16, ring clouds): how many groups a mask
order has to form, and how the rows are distributed over group sizes."""
import os, sys
import torch
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
    out = m(bd)
    idict = out["encoded_spconv_tensor"].indice_dict
    for key, d in idict.items():
        if key == "__grid_token__":
            continue
        for name in ("nbr", "nbr_t"):
            t = d[name]
            if name == "nbr_t" and d["subm"]:
                continue
            if t.shape[1] > 31:
                continue
            masks, _ = ops.mask_order(t)
            u, cnt = torch.unique(masks, return_counts=True)
            n = t.shape[0]
            big128 = float(cnt[cnt >= 128].sum()) / n
            big32 = float(cnt[cnt >= 32].sum()) / n
            print(f"{key:13s} {name:5s} rows {n:7d} K {t.shape[1]:2d} distinct masks {u.numel():7d}  rows in groups >=128: {big128*100:5.1f}%  >=32: {big32*100:5.1f}%  "
                  f"largest group {int(cnt.max())}", flush=True)
