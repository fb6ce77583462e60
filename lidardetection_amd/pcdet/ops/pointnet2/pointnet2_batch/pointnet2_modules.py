"""Dense-batch set abstraction / feature propagation modules with the reference's class names, keyword arguments and parameter
layout (pcdet/ops/pointnet2/pointnet2_batch/pointnet2_modules.py: PointnetSAModuleMSG :61-101, PointnetSAModule :104-121,
PointnetFPModule :124-170)."""
from typing import List

import torch
import torch.nn as nn

from . import pointnet2_utils
from .. import _common as C


class _PointnetSAModuleBase(nn.Module):
    def __init__(self):
        super().__init__()
        self.npoint, self.groupers, self.mlps, self.pool_method = None, None, None, 'max_pool'

    def _sample_centres(self, xyz):
        if self.npoint is None:
            return None
        picked = pointnet2_utils.furthest_point_sample(xyz, self.npoint)
        return pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), picked).transpose(1, 2).contiguous()

    def forward(self, xyz, features=None, new_xyz=None):
        """xyz (B, N, 3), features (B, C, N) -> (new_xyz (B, npoint, 3), (B, sum of the scales' last MLP widths, npoint))"""
        if new_xyz is None:
            new_xyz = self._sample_centres(xyz)
        per_scale = [C.pool_over_samples(mlp(grouper(xyz, new_xyz, features)), self.pool_method)
                     for grouper, mlp in zip(self.groupers, self.mlps)]
        return new_xyz, torch.cat(per_scale, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    def __init__(self, *, npoint: int, radii: List[float], nsamples: List[int], mlps: List[List[int]], bn: bool = True,
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__()
        if not (len(radii) == len(nsamples) == len(mlps)):
            raise AssertionError('one radius, sample count and MLP per scale')
        self.npoint, self.pool_method = npoint, pool_method
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, widths in zip(radii, nsamples, mlps):
            grouper = (pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz) if npoint is not None
                       else pointnet2_utils.GroupAll(use_xyz))
            self.groupers.append(grouper)
            if use_xyz:
                widths[0] += 3        # modifies the caller's list, as the reference does
            self.mlps.append(C.shared_mlp(widths))


class PointnetSAModule(PointnetSAModuleMSG):
    """single-scale convenience form"""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None, nsample: int = None, bn: bool = True,
                 use_xyz: bool = True, pool_method='max_pool'):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz,
                         pool_method=pool_method)


class PointnetFPModule(nn.Module):
    def __init__(self, *, mlp: List[int], bn: bool = True):
        super().__init__()
        self.mlp = C.shared_mlp(mlp)

    def forward(self, unknown, known, unknow_feats, known_feats):
        """unknown (B, n, 3), known (B, m, 3) or None, unknow_feats (B, C1, n) or None, known_feats (B, C2, m) -> (B, mlp[-1], n)"""
        if known is None:
            y = known_feats.expand(known_feats.shape[0], known_feats.shape[1], unknown.shape[1])
        else:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            y = pointnet2_utils.three_interpolate(known_feats, idx, C.inverse_distance_weights(dist))
        if unknow_feats is not None:
            y = torch.cat((y, unknow_feats), dim=1)
        return self.mlp(y.unsqueeze(-1)).squeeze(-1)
