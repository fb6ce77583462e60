"""torch-CPU / numpy restatement of the reference's pure-torch PointPillar / SECOND front-end modules.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned against golden vectors emitted by the
reference's own modules (tests/golden/make_golden.py -> tests/golden/pp_modules_*.npz).

  pillar_vfe        /root/reference/pcdet/models/backbones_3d/vfe/pillar_vfe.py:29-49, 86-123
  mean_vfe          /root/reference/pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31
  pillar_scatter    /root/reference/pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37
  collate           /root/reference/pcdet/datasets/dataset.py:153-185 (voxels/voxel_coords part)
"""
import numpy as np
import torch


def collate(per_frame):
    """per_frame: list of (voxels, coords(z,y,x), num) -> voxels (SV,P,C), coords (SV,4) [b,z,y,x], num (SV,)."""
    vox = np.concatenate([f[0] for f in per_frame], 0)
    num = np.concatenate([f[2] for f in per_frame], 0)
    coords = np.concatenate([np.pad(f[1], ((0, 0), (1, 0)), mode="constant", constant_values=i)
                             for i, f in enumerate(per_frame)], 0)
    return vox, coords, num


def pillar_vfe(voxels, num_points, coords, weight, bn_gamma, bn_beta, bn_mean, bn_var, voxel_size, pc_range,
               eps=1e-3, use_absolute_xyz=True, with_distance=False):
    """Eval-mode PillarVFE with one PFN layer (the PointPillar-KITTI config).  All args torch CPU f32.
    coords are (V,4) [b,z,y,x] as float (models/__init__.py:22 casts them)."""
    vf = voxels
    vx, vy, vz = voxel_size
    xo, yo, zo = vx / 2 + pc_range[0], vy / 2 + pc_range[1], vz / 2 + pc_range[2]
    points_mean = vf[:, :, :3].sum(dim=1, keepdim=True) / num_points.type_as(vf).view(-1, 1, 1)
    f_cluster = vf[:, :, :3] - points_mean
    f_center = torch.zeros_like(vf[:, :, :3])
    f_center[:, :, 0] = vf[:, :, 0] - (coords[:, 3].to(vf.dtype).unsqueeze(1) * vx + xo)
    f_center[:, :, 1] = vf[:, :, 1] - (coords[:, 2].to(vf.dtype).unsqueeze(1) * vy + yo)
    f_center[:, :, 2] = vf[:, :, 2] - (coords[:, 1].to(vf.dtype).unsqueeze(1) * vz + zo)
    feats = [vf, f_cluster, f_center] if use_absolute_xyz else [vf[..., 3:], f_cluster, f_center]
    if with_distance:
        feats.append(torch.norm(vf[:, :, :3], 2, 2, keepdim=True))
    feats = torch.cat(feats, dim=-1)
    P = feats.shape[1]
    mask = (num_points.int().unsqueeze(1) > torch.arange(P, dtype=torch.int).view(1, -1)).unsqueeze(-1).type_as(vf)
    feats = feats * mask
    x = feats @ weight.t()                                   # Linear(10->64, bias=False)
    x = (x - bn_mean) / torch.sqrt(bn_var + eps) * bn_gamma + bn_beta   # BatchNorm1d eval
    x = torch.relu(x)
    return x.max(dim=1)[0]


def mean_vfe(voxels, num_points):
    s = voxels.sum(dim=1)
    return (s / torch.clamp_min(num_points.view(-1, 1), min=1.0).type_as(voxels)).contiguous()


def pillar_scatter(pillar_features, coords, batch_size, nx, ny, nz=1):
    """-> (B, C*nz, ny, nx).  coords (V,4) [b,z,y,x]."""
    C = pillar_features.shape[1]
    out = torch.zeros(batch_size, C, nz * nx * ny, dtype=pillar_features.dtype)
    for b in range(batch_size):
        m = coords[:, 0] == b
        tc = coords[m]
        idx = (tc[:, 1] + tc[:, 2] * nx + tc[:, 3]).long()
        out[b][:, idx] = pillar_features[m].t()
    return out.view(batch_size, C * nz, ny, nx)
