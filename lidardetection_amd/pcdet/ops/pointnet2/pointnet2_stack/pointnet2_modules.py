"""Mirror of pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py: StackSAModuleMSG (:10-92) and
StackPointnetFPModule (:95-137).  Parameter names / shapes match the reference so checkpoints load."""
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


class StackSAModuleMSG(nn.Module):
    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 pool_method='max_pool'):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            if use_xyz:
                mlp_spec[0] += 3      # in place, like the reference (callers rely on the side effect)
            self.mlps.append(_shared_mlp(mlp_spec))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        """xyz (N,3), new_xyz (M,3), features (N,C) -> (new_xyz, new_features (M, sum_k mlps[k][-1]))."""
        outs = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped, _ = grouper(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)     # (M, C, nsample)
            x = mlp(grouped.permute(1, 0, 2).unsqueeze(dim=0))                                 # (1, C', M, nsample)
            if self.pool_method == 'max_pool':
                x = F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(dim=-1)
            elif self.pool_method == 'avg_pool':
                x = F.avg_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(dim=-1)
            else:
                raise NotImplementedError
            outs.append(x.squeeze(dim=0).permute(1, 0))                                        # (M, C')
        return new_xyz, torch.cat(outs, dim=1)


class StackPointnetFPModule(nn.Module):
    def __init__(self, *, mlp: List[int]):
        super().__init__()
        self.mlp = _shared_mlp(mlp)

    def forward(self, unknown, unknown_batch_cnt, known, known_batch_cnt, unknown_feats=None, known_feats=None):
        """inverse-distance interpolation of known_feats (M,C2) onto unknown (N,3) + shared MLP -> (N, C_out)."""
        dist, idx = pointnet2_utils.three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        dist_recip = 1.0 / (dist + 1e-8)
        weight = dist_recip / torch.sum(dist_recip, dim=-1, keepdim=True)
        x = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        if unknown_feats is not None:
            x = torch.cat([x, unknown_feats], dim=1)
        x = self.mlp(x.permute(1, 0)[None, :, :, None])
        return x.squeeze(dim=0).squeeze(dim=-1).permute(1, 0)
