"""Mirror of pcdet/models/backbones_2d/map_to_bev/height_compression.py:10-26 (dense() is one write-once HIP pass)."""
import torch.nn as nn


class HeightCompression(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES

    def forward(self, batch_dict):
        dense = batch_dict['encoded_spconv_tensor'].dense()
        N, C, D, H, W = dense.shape
        batch_dict['spatial_features'] = dense.view(N, C * D, H, W)
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict
