"""Mirror of pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31 on the HIP mean kernel (torch ops when gradients are needed)."""
import torch

from ..... import pillar_ops
from .vfe_template import VFETemplate


class MeanVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        """voxels (V, P, C), voxel_num_points (V,) -> voxel_features (V, C) = sum over P / max(count, 1)."""
        voxels, num = batch_dict['voxels'], batch_dict['voxel_num_points']
        if voxels.requires_grad or not voxels.is_cuda:
            s = voxels.sum(dim=1, keepdim=False)
            batch_dict['voxel_features'] = (s / torch.clamp_min(num.view(-1, 1), min=1.0).type_as(voxels)).contiguous()
        else:
            n = num if num.dtype in (torch.int32, torch.float32) else num.float()
            batch_dict['voxel_features'] = pillar_ops.mean_vfe(voxels.contiguous(), n.contiguous())
        return batch_dict
