"""Mirror of pcdet/models/backbones_3d/vfe/pillar_vfe.py (PFNLayer :8-49, PillarVFE :52-123).

Eval mode with the standard single-PFN-layer config runs the fused HIP kernel (mean / cluster / centre augmentation,
Linear, folded BatchNorm, ReLU, max over the pillar's points in one pass over the occupied point slots).  Training,
multi-layer PFN stacks and USE_ABSLOTE_XYZ=False run the reference's op sequence on stock torch."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..... import pillar_ops
from .vfe_template import VFETemplate


class PFNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe, self.use_norm = last_layer, use_norm
        if not self.last_vfe:
            out_channels = out_channels // 2
        if self.use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)
        self.part = 50000

    def forward(self, inputs):
        x = self.linear(inputs)
        x = self.norm(x.permute(0, 2, 1)).permute(0, 2, 1) if self.use_norm else x
        x = F.relu(x)
        x_max = torch.max(x, dim=1, keepdim=True)[0]
        if self.last_vfe:
            return x_max
        return torch.cat([x, x_max.repeat(1, inputs.shape[1], 1)], dim=2)

    def folded(self):
        """(weight (cout, cin), scale, shift) with BatchNorm (eval) folded; no-norm: scale 1, shift = bias."""
        w = self.linear.weight.detach().contiguous()
        if self.use_norm:
            n = self.norm
            s, t = pillar_ops.fold_bn(n.weight.detach(), n.bias.detach(), n.running_mean, n.running_var, n.eps)
            return w, s, t
        return w, torch.ones_like(self.linear.bias), self.linear.bias.detach().contiguous()


class PillarVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range):
        super().__init__(model_cfg=model_cfg)
        self.use_norm = self.model_cfg.USE_NORM
        self.with_distance = self.model_cfg.WITH_DISTANCE
        self.use_absolute_xyz = self.model_cfg.USE_ABSLOTE_XYZ
        self.raw_point_features = num_point_features
        num_point_features += 6 if self.use_absolute_xyz else 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = self.model_cfg.NUM_FILTERS
        assert len(self.num_filters) > 0
        num_filters = [num_point_features] + list(self.num_filters)
        self.pfn_layers = nn.ModuleList([
            PFNLayer(num_filters[i], num_filters[i + 1], self.use_norm, last_layer=(i >= len(num_filters) - 2))
            for i in range(len(num_filters) - 1)])
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.voxel_x, self.voxel_y, self.voxel_z = self.voxel_size
        self.x_offset = self.voxel_x / 2 + point_cloud_range[0]
        self.y_offset = self.voxel_y / 2 + point_cloud_range[1]
        self.z_offset = self.voxel_z / 2 + point_cloud_range[2]

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def get_paddings_indicator(self, actual_num, max_num, axis=0):
        actual_num = torch.unsqueeze(actual_num, axis + 1)
        shape = [1] * len(actual_num.shape)
        shape[axis + 1] = -1
        return actual_num.int() > torch.arange(max_num, dtype=torch.int, device=actual_num.device).view(shape)

    def _fused_ok(self, voxels):
        return (not self.training and voxels.is_cuda and not voxels.requires_grad and len(self.pfn_layers) == 1
                and self.use_absolute_xyz and voxels.shape[1] <= 64 and self.num_filters[-1] <= 64
                and 3 <= self.raw_point_features <= 8)

    def forward(self, batch_dict, **kwargs):
        vf, num, coords = batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords']
        if self._fused_ok(vf):
            w, s, t = self.pfn_layers[0].folded()
            cast = lambda a: a if a.dtype in (torch.int32, torch.float32) else a.float()
            batch_dict['pillar_features'] = pillar_ops.pillar_vfe(vf.contiguous(), cast(num).contiguous(), cast(coords).contiguous(),
                                                                  w, s, t, self.voxel_size, self.point_cloud_range,
                                                                  with_distance=self.with_distance)
            return batch_dict
        points_mean = vf[:, :, :3].sum(dim=1, keepdim=True) / num.type_as(vf).view(-1, 1, 1)
        f_cluster = vf[:, :, :3] - points_mean
        f_center = torch.zeros_like(vf[:, :, :3])
        f_center[:, :, 0] = vf[:, :, 0] - (coords[:, 3].to(vf.dtype).unsqueeze(1) * self.voxel_x + self.x_offset)
        f_center[:, :, 1] = vf[:, :, 1] - (coords[:, 2].to(vf.dtype).unsqueeze(1) * self.voxel_y + self.y_offset)
        f_center[:, :, 2] = vf[:, :, 2] - (coords[:, 1].to(vf.dtype).unsqueeze(1) * self.voxel_z + self.z_offset)
        feats = [vf, f_cluster, f_center] if self.use_absolute_xyz else [vf[..., 3:], f_cluster, f_center]
        if self.with_distance:
            feats.append(torch.norm(vf[:, :, :3], 2, 2, keepdim=True))
        feats = torch.cat(feats, dim=-1)
        mask = self.get_paddings_indicator(num, feats.shape[1], axis=0)
        feats = feats * torch.unsqueeze(mask, -1).type_as(vf)
        for pfn in self.pfn_layers:
            feats = pfn(feats)
        batch_dict['pillar_features'] = feats.squeeze()
        return batch_dict
