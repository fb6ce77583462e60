"""Point-cloud RoI pooling under the reference's public names (pcdet/ops/roipoint_pool3d/roipoint_pool3d_utils.py:9-59),
bound to lidardetection_amd.ext.roipoint_pool3d_cuda: for every (enlarged) box, the first `num_sampled_points` points
inside it, cycled when there are fewer, with their features; boxes without points are flagged.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from ...utils import box_utils
from ....ext import roipoint_pool3d_cuda as _native


class RoIPointPool3dFunction(Function):
    @staticmethod
    def forward(ctx, points, point_features, boxes3d, pool_extra_width, num_sampled_points=512):
        """points (B, N, 3), point_features (B, N, C), boxes3d (B, M, 7)
        -> pooled (B, M, num_sampled_points, 3 + C), empty flag (B, M) int32"""
        if points.dim() != 3 or points.shape[2] != 3:
            raise AssertionError('points must be (B, N, 3)')
        n_batch, n_box, n_feat = points.shape[0], boxes3d.shape[1], point_features.shape[2]
        grown = box_utils.enlarge_box3d(boxes3d.reshape(-1, 7), pool_extra_width).reshape(n_batch, n_box, 7)
        pooled = point_features.new_zeros((n_batch, n_box, num_sampled_points, 3 + n_feat))
        empty = torch.zeros((n_batch, n_box), dtype=torch.int32, device=point_features.device)
        _native.forward(points.contiguous(), grown.contiguous(), point_features.contiguous(), pooled, empty)
        return pooled, empty

    @staticmethod
    def backward(ctx, grad_out):
        raise NotImplementedError    # the reference defines no gradient for this op either


class RoIPointPool3d(nn.Module):
    def __init__(self, num_sampled_points=512, pool_extra_width=1.0):
        super().__init__()
        self.num_sampled_points, self.pool_extra_width = num_sampled_points, pool_extra_width

    def forward(self, points, point_features, boxes3d):
        return RoIPointPool3dFunction.apply(points, point_features, boxes3d, self.pool_extra_width, self.num_sampled_points)
