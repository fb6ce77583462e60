"""Mirror of pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37 on the write-once HIP scatter
(no `.item()` host sync when batch_dict carries 'batch_size', no per-sample Python loop, no zeros+assign+stack passes)."""
import torch
import torch.nn as nn

from ..... import pillar_ops


class PointPillarScatter(nn.Module):
    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES
        self.nx, self.ny, self.nz = [int(v) for v in grid_size]
        assert self.nz == 1

    def forward(self, batch_dict, **kwargs):
        feats, coords = batch_dict['pillar_features'], batch_dict['voxel_coords']
        batch_size = batch_dict['batch_size'] if 'batch_size' in batch_dict else coords[:, 0].max().int().item() + 1
        if feats.is_cuda and not feats.requires_grad and feats.shape[1] in (32, 64, 128):
            c = coords if coords.dtype in (torch.int32, torch.float32) else coords.float()
            batch_dict['spatial_features'] = pillar_ops.pillar_scatter(feats.contiguous(), c.contiguous(), batch_size, self.nx, self.ny)
            return batch_dict
        out = feats.new_zeros((batch_size, self.num_bev_features, self.nz * self.nx * self.ny))
        idx = (coords[:, 1] + coords[:, 2] * self.nx + coords[:, 3]).long()
        out[coords[:, 0].long(), :, idx] = feats
        batch_dict['spatial_features'] = out.view(batch_size, self.num_bev_features * self.nz, self.ny, self.nx)
        return batch_dict
