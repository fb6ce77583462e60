"""Mirror of pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py (stacked-batch PointNet++ operators).

Public symbols and call signatures are the reference's: ball_query, grouping_operation, QueryAndGroup,
furthest_point_sample, three_nn, three_interpolate.  Native module: lidardetection_amd.ext.pointnet2_stack_cuda.
Layout reminder: "stacked" tensors concatenate the samples of a batch along dim 0 and come with a
(batch_size,) int32 count tensor.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from .....ext import pointnet2_stack_cuda as pointnet2


def _i32(shape, device, zero=False):
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.int32, device=device)


def _f32(shape, device, zero=False):
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=device)


class BallQuery(Function):
    """pointnet2_utils.py:8-43 — returns (idx (M, nsample) int32, empty_ball_mask (M,) bool)."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt):
        for t in (new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt):
            assert t.is_contiguous()
        B, M = xyz_batch_cnt.shape[0], new_xyz.shape[0]
        idx = _i32((M, nsample), new_xyz.device, zero=True)
        pointnet2.ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
        empty_ball_mask = idx[:, 0] == -1          # the kernel's sentinel for "nothing within the radius"
        idx[empty_ball_mask] = 0
        return idx, empty_ball_mask

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    """pointnet2_utils.py:48-105 — features (N, C) + idx (M, nsample) -> (M, C, nsample); differentiable."""

    @staticmethod
    def forward(ctx, features, features_batch_cnt, idx, idx_batch_cnt):
        for t in (features, features_batch_cnt, idx, idx_batch_cnt):
            assert t.is_contiguous()
        assert features.shape[0] == features_batch_cnt.sum(), \
            'features: %s, features_batch_cnt: %s' % (str(features.shape), str(features_batch_cnt))
        assert idx.shape[0] == idx_batch_cnt.sum(), 'idx: %s, idx_batch_cnt: %s' % (str(idx.shape), str(idx_batch_cnt))
        (M, nsample), (N, C) = idx.size(), features.size()
        B = idx_batch_cnt.shape[0]
        output = _f32((M, C, nsample), features.device)
        pointnet2.group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, output)
        ctx.for_backwards = (B, N, idx, features_batch_cnt, idx_batch_cnt)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        B, N, idx, features_batch_cnt, idx_batch_cnt = ctx.for_backwards
        M, C, nsample = grad_out.size()
        grad_features = _f32((N, C), grad_out.device, zero=True)
        pointnet2.group_points_grad_wrapper(B, M, C, N, nsample, grad_out.detach().contiguous(), idx, idx_batch_cnt,
                                            features_batch_cnt, grad_features)
        return grad_features, None, None, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(nn.Module):
    """pointnet2_utils.py:108-155 — ball query + grouping of (relative xyz | features)."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        assert xyz.shape[0] == xyz_batch_cnt.sum(), 'xyz: %s, xyz_batch_cnt: %s' % (str(xyz.shape), str(new_xyz_batch_cnt))
        assert new_xyz.shape[0] == new_xyz_batch_cnt.sum(), \
            'new_xyz: %s, new_xyz_batch_cnt: %s' % (str(new_xyz.shape), str(new_xyz_batch_cnt))
        idx, empty_ball_mask = ball_query(self.radius, self.nsample, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        grouped_xyz = grouping_operation(xyz, xyz_batch_cnt, idx, new_xyz_batch_cnt)   # (M, 3, nsample)
        grouped_xyz -= new_xyz.unsqueeze(-1)
        grouped_xyz[empty_ball_mask] = 0
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz, idx
        grouped_features = grouping_operation(features, xyz_batch_cnt, idx, new_xyz_batch_cnt)  # (M, C, nsample)
        grouped_features[empty_ball_mask] = 0
        new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz else grouped_features
        return new_features, idx


class FurthestPointSampling(Function):
    """pointnet2_utils.py:158-184 — xyz (B, N, 3) -> (B, npoint) int32, starting from point 0."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        assert xyz.is_contiguous()
        B, N, _ = xyz.size()
        output = _i32((B, npoint), xyz.device)
        temp = torch.full((B, N), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.furthest_point_sampling_wrapper(B, N, npoint, xyz, temp, output)
        return output

    @staticmethod
    def backward(xyz, a=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class ThreeNN(Function):
    """pointnet2_utils.py:187-220 — 3 nearest known points: (dist (N,3) = sqrt(d2), idx (N,3) global int32)."""

    @staticmethod
    def forward(ctx, unknown, unknown_batch_cnt, known, known_batch_cnt):
        assert unknown.shape.__len__() == 2 and unknown.shape[1] == 3
        assert known.shape.__len__() == 2 and known.shape[1] == 3
        assert unknown_batch_cnt.__len__() == known_batch_cnt.__len__()
        dist2 = unknown.new_zeros(unknown.shape)
        idx = unknown_batch_cnt.new_zeros(unknown.shape).int()
        pointnet2.three_nn_wrapper(unknown.contiguous(), unknown_batch_cnt.contiguous(), known.contiguous(),
                                   known_batch_cnt.contiguous(), dist2, idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None, None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """pointnet2_utils.py:223-262 — out[n] = sum_k weight[n,k] * features[idx[n,k]]; differentiable in features."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        assert idx.shape[0] == weight.shape[0] and idx.shape[1] == weight.shape[1] == 3
        ctx.three_interpolate_for_backward = (idx, weight, features.shape[0])
        output = features.new_zeros((idx.shape[0], features.shape[1]))
        pointnet2.three_interpolate_wrapper(features.contiguous(), idx.contiguous(), weight.contiguous(), output)
        return output

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, M = ctx.three_interpolate_for_backward
        grad_features = grad_out.new_zeros((M, grad_out.shape[1]))
        pointnet2.three_interpolate_grad_wrapper(grad_out.contiguous(), idx.contiguous(), weight.contiguous(), grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply
