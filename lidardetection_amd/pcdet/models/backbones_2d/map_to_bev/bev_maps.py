"""Sparse -> dense BEV map modules of the reference, same class names / config keys / batch_dict keys:
PointPillarScatter (pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37) and HeightCompression
(height_compression.py:10-26), on this repo's write-once HIP scatter (`pillar_ops.pillar_scatter`, `SparseConvTensor.dense`).
"""
import torch
import torch.nn as nn

from ..... import pillar_ops


class PointPillarScatter(nn.Module):
    """pillar_features (V, C) at voxel_coords (V, 4) [b, z, y, x] -> spatial_features (B, C, ny, nx); nz must be 1"""

    def __init__(self, model_cfg, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg.NUM_BEV_FEATURES
        self.nx, self.ny, self.nz = (int(v) for v in grid_size)
        if self.nz != 1:
            raise AssertionError('PointPillarScatter expects a single z slice')

    def _batch_size(self, batch_dict, coords):
        if 'batch_size' in batch_dict:                       # no device read-back when the pipeline carries it
            return batch_dict['batch_size']
        return int(coords[:, 0].max()) + 1

    def forward(self, batch_dict, **kwargs):
        feats, coords = batch_dict['pillar_features'], batch_dict['voxel_coords']
        n_batch = self._batch_size(batch_dict, coords)
        if feats.is_cuda and not feats.requires_grad and feats.shape[1] in (32, 64, 128):
            c = coords if coords.dtype in (torch.int32, torch.float32) else coords.float()
            canvas = pillar_ops.pillar_scatter(feats.contiguous(), c.contiguous(), n_batch, self.nx, self.ny)
        else:                                                # differentiable / odd-width path on stock torch
            flat = feats.new_zeros((n_batch, self.num_bev_features, self.ny * self.nx))
            cell = (coords[:, 2] * self.nx + coords[:, 3] + coords[:, 1]).long()
            flat[coords[:, 0].long(), :, cell] = feats
            canvas = flat.view(n_batch, self.num_bev_features * self.nz, self.ny, self.nx)
        batch_dict['spatial_features'] = canvas
        return batch_dict


class HeightCompression(nn.Module):
    """encoded_spconv_tensor -> spatial_features (B, C * D, H, W): the depth axis folded into the channels"""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = model_cfg.NUM_BEV_FEATURES

    def forward(self, batch_dict):
        volume = batch_dict['encoded_spconv_tensor'].dense()            # (B, C, D, H, W), every element written once
        b, c, d, h, w = volume.shape
        batch_dict['spatial_features'] = volume.view(b, c * d, h, w)
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict
