"""Voxel feature encoders of the reference (pcdet/models/backbones_3d/vfe/: VFETemplate, MeanVFE mean_vfe.py:14-31, PFNLayer /
PillarVFE pillar_vfe.py:8-123) with the reference's class names, constructor arguments, batch_dict keys and parameter names
(`pfn_layers.N.linear / .norm`, so its checkpoints load), built around this repo's HIP kernels:

  * PillarVFE in eval mode with one PFN layer -> `pillar_ops.pillar_vfe`: decoration (cluster / centre offsets, optional
    range), Linear, folded BatchNorm, ReLU and the max over a pillar's points in one pass over the occupied slots;
  * MeanVFE -> `pillar_ops.mean_vfe`.
Training (gradients), stacked PFN layers and USE_ABSLOTE_XYZ=False take the plain-torch formulation below.
"""
import torch
import torch.nn as nn

from ..... import pillar_ops


def _as_kernel_dtype(t):
    """counts / coordinates reach the kernels as int32 or float32 (whatever the pipeline produced, without a copy when possible)"""
    return t if t.dtype in (torch.int32, torch.float32) else t.float()


class VFETemplate(nn.Module):
    """common interface: keeps the config, reports the width of the features it emits"""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg

    def get_output_feature_dim(self):
        raise NotImplementedError

    def forward(self, **kwargs):
        raise NotImplementedError


class MeanVFE(VFETemplate):
    """voxels (V, P, C) + voxel_num_points (V,) -> voxel_features (V, C): mean over the real points of each voxel"""

    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        voxels, counts = batch_dict['voxels'], batch_dict['voxel_num_points']
        if voxels.is_cuda and not voxels.requires_grad:
            mean = pillar_ops.mean_vfe(voxels.contiguous(), _as_kernel_dtype(counts).contiguous())
        else:   # padded slots are zero, so the row sum is the sum over the real points
            mean = (voxels.sum(dim=1) / counts.reshape(-1, 1).clamp(min=1.0).to(voxels.dtype)).contiguous()
        batch_dict['voxel_features'] = mean
        return batch_dict


class PFNLayer(nn.Module):
    """Linear (+ BatchNorm1d) + ReLU per point, max over the pillar; non-final layers append the max to every point."""

    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe, self.use_norm = last_layer, use_norm
        width = out_channels if last_layer else out_channels // 2
        self.linear = nn.Linear(in_channels, width, bias=not use_norm)
        if use_norm:
            self.norm = nn.BatchNorm1d(width, eps=1e-3, momentum=0.01)
        self.part = 50000       # kept for attribute compatibility with the reference

    def forward(self, inputs):
        y = self.linear(inputs)
        if self.use_norm:
            y = self.norm(y.transpose(1, 2)).transpose(1, 2)
        y = torch.relu(y)
        pooled = y.amax(dim=1, keepdim=True)
        if self.last_vfe:
            return pooled
        return torch.cat((y, pooled.expand(-1, y.shape[1], -1)), dim=2)

    def folded(self):
        """(weight (cout, cin), per-channel scale, shift): eval-mode BatchNorm folded; without a norm: scale 1, shift = bias"""
        weight = self.linear.weight.detach().contiguous()
        if not self.use_norm:
            return weight, torch.ones_like(self.linear.bias), self.linear.bias.detach().contiguous()
        bn = self.norm
        scale, shift = pillar_ops.fold_bn(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        return weight, scale, shift


class PillarVFE(VFETemplate):
    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range):
        super().__init__(model_cfg=model_cfg)
        cfg = self.model_cfg
        self.use_norm, self.with_distance, self.use_absolute_xyz = cfg.USE_NORM, cfg.WITH_DISTANCE, cfg.USE_ABSLOTE_XYZ
        self.raw_point_features = num_point_features
        self.num_filters = list(cfg.NUM_FILTERS)
        if not self.num_filters:
            raise AssertionError('NUM_FILTERS must name at least one PFN layer')
        decorated = num_point_features + (6 if self.use_absolute_xyz else 3) + (1 if self.with_distance else 0)
        widths = [decorated] + self.num_filters
        last = len(widths) - 2
        self.pfn_layers = nn.ModuleList(PFNLayer(widths[i], widths[i + 1], self.use_norm, last_layer=(i >= last))
                                        for i in range(len(widths) - 1))
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.voxel_x, self.voxel_y, self.voxel_z = self.voxel_size
        self.x_offset, self.y_offset, self.z_offset = (self.voxel_size[a] / 2 + self.point_cloud_range[a] for a in range(3))

    def get_output_feature_dim(self):
        return self.num_filters[-1]

    def get_paddings_indicator(self, actual_num, max_num, axis=0):
        """True for the real point slots of each voxel (reference helper, same name and arguments)"""
        slots = torch.arange(max_num, dtype=torch.int32, device=actual_num.device)
        shape = [1] * (actual_num.dim() + 1)
        shape[axis + 1] = -1
        return actual_num.int().unsqueeze(axis + 1) > slots.view(shape)

    def _kernel_path(self, voxels):
        return (not self.training and voxels.is_cuda and not voxels.requires_grad and len(self.pfn_layers) == 1
                and self.use_absolute_xyz and voxels.shape[1] <= 64 and self.num_filters[-1] <= 64
                and 3 <= self.raw_point_features <= 8)

    def _decorate(self, voxels, counts, coords):
        """per-point inputs of the first PFN layer: raw features, offset to the pillar's point mean, offset to its centre"""
        xyz = voxels[:, :, :3]
        mean = xyz.sum(dim=1, keepdim=True) / counts.to(voxels.dtype).view(-1, 1, 1)
        cell = coords[:, [3, 2, 1]].to(voxels.dtype)                    # (x, y, z) cell of every pillar
        centre = cell * voxels.new_tensor(self.voxel_size) + voxels.new_tensor([self.x_offset, self.y_offset, self.z_offset])
        parts = [voxels if self.use_absolute_xyz else voxels[..., 3:], xyz - mean, xyz - centre.unsqueeze(1)]
        if self.with_distance:
            parts.append(xyz.norm(dim=2, keepdim=True))
        real = self.get_paddings_indicator(counts, voxels.shape[1], axis=0).unsqueeze(-1)
        return torch.cat(parts, dim=-1) * real.to(voxels.dtype)

    def forward(self, batch_dict, **kwargs):
        voxels, counts, coords = batch_dict['voxels'], batch_dict['voxel_num_points'], batch_dict['voxel_coords']
        if self._kernel_path(voxels):
            weight, scale, shift = self.pfn_layers[0].folded()
            batch_dict['pillar_features'] = pillar_ops.pillar_vfe(
                voxels.contiguous(), _as_kernel_dtype(counts).contiguous(), _as_kernel_dtype(coords).contiguous(), weight, scale,
                shift, self.voxel_size, self.point_cloud_range, with_distance=self.with_distance)
            return batch_dict
        feats = self._decorate(voxels, counts, coords)
        for layer in self.pfn_layers:
            feats = layer(feats)
        batch_dict['pillar_features'] = feats.squeeze()
        return batch_dict
