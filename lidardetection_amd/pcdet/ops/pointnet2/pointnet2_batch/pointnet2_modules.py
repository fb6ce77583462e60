"""Mirror of pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py: PointnetSAModuleMSG (:61-101),
PointnetSAModule (:104-121), PointnetFPModule (:124-170).  Parameter names / shapes match the reference."""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils


def _shared_mlp(spec):
    layers = []
    for cin, cout in zip(spec[:-1], spec[1:]):
        layers += [nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()]
    return nn.Sequential(*layers)


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint, self.groupers, self.mlps, self.pool_method = None, None, None, 'max_pool'

    def forward(self, xyz, features=None, new_xyz=None):
        """xyz (B, N, 3), features (B, C, N) -> new_xyz (B, npoint, 3), new_features (B, sum_k mlps[k][-1], npoint)."""
        if new_xyz is None:
            new_xyz = pointnet2_utils.gather_operation(
                xyz.transpose(1, 2).contiguous(), pointnet2_utils.furthest_point_sample(xyz, self.npoint)
            ).transpose(1, 2).contiguous() if self.npoint is not None else None
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            x = mlp(grouper(xyz, new_xyz, features))                                   # (B, mlp[-1], npoint, nsample)
            if self.pool_method == 'max_pool':
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)])
            elif self.pool_method == 'avg_pool':
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)])
            else:
                raise NotImplementedError
            outs.append(x.squeeze(-1))
        return new_xyz, torch.cat(outs, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]], bn: bool = True,
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz)
                                 if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            self.mlps.append(_shared_mlp(mlp_spec))
        self.pool_method = pool_method


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None, bn: bool = True,
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method)


class PointnetFPModule(nn.Module):
    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = _shared_mlp(mlp)

    def forward(self, unknown, known, unknow_feats, known_feats):
        """unknown (B,n,3), known (B,m,3), unknow_feats (B,C1,n), known_feats (B,C2,m) -> (B, mlp[-1], n)."""
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            x = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            x = known_feats.expand(*known_feats.size()[0:2], unknown.size(1))
        if unknow_feats is not None:
            x = torch.cat([x, unknow_feats], dim=1)
        return self.mlp(x.unsqueeze(-1)).squeeze(-1)
