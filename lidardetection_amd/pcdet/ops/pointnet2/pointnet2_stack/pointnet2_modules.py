"""Stacked-batch set abstraction / feature propagation modules with the reference's class names, keyword arguments and
parameter layout (pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py: StackSAModuleMSG :10-92, StackPointnetFPModule
:95-137), so its checkpoints load."""
from typing import List

import torch
import torch.nn as nn

from . import pointnet2_utils
from .. import _common as C


class StackSAModuleMSG(nn.Module):
    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 pool_method='max_pool'):
        super().__init__()
        if not (len(radii) == len(nsamples) == len(mlps)):
            raise AssertionError('one radius, sample count and MLP per scale')
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, widths in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            if use_xyz:
                widths[0] += 3        # modifies the caller's list, as the reference does (callers read it back)
            self.mlps.append(C.shared_mlp(widths))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for layer in self.modules():
            if isinstance(layer, nn.Conv2d):
                nn.init.kaiming_normal_(layer.weight)
                if layer.bias is not None:
                    nn.init.zeros_(layer.bias)
            elif isinstance(layer, nn.BatchNorm2d):
                nn.init.ones_(layer.weight)
                nn.init.zeros_(layer.bias)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        """xyz (N, 3), features (N, C), new_xyz (M, 3) -> (new_xyz, (M, sum of the scales' last MLP widths))"""
        per_scale = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped, _ = grouper(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)      # (M, C, nsample)
            y = mlp(grouped.permute(1, 0, 2).unsqueeze(0))                                      # (1, C', M, nsample)
            per_scale.append(C.pool_over_samples(y, self.pool_method)[0].t())                    # (M, C')
        return new_xyz, torch.cat(per_scale, dim=1)


class StackPointnetFPModule(nn.Module):
    def __init__(self, *, mlp: List[int]):
        super().__init__()
        self.mlp = C.shared_mlp(mlp)

    def forward(self, unknown, unknown_batch_cnt, known, known_batch_cnt, unknown_feats=None, known_feats=None):
        """known_feats (M, C2) interpolated onto unknown (N, 3) by inverse distance to the 3 nearest known points,
        concatenated with unknown_feats, then the shared MLP -> (N, C_out)"""
        dist, idx = pointnet2_utils.three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        y = pointnet2_utils.three_interpolate(known_feats, idx, C.inverse_distance_weights(dist))
        if unknown_feats is not None:
            y = torch.cat((y, unknown_feats), dim=1)
        return self.mlp(y.t()[None, :, :, None])[0, :, :, 0].t()
