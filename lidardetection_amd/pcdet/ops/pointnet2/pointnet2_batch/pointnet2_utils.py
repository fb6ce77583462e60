"""Dense-batch PointNet++ operators under the reference's public names (pcdet/ops/pointnet2/pointnet2_batch/
pointnet2_utils.py:10-290): furthest_point_sample, gather_operation, three_nn, three_interpolate, grouping_operation,
ball_query, QueryAndGroup, GroupAll.  Coordinates are (B, N, 3), features channel-major (B, C, N).
Native module: lidardetection_amd.ext.pointnet2_batch_cuda."""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _common as C
from .....ext import pointnet2_batch_cuda as pointnet2


class FurthestPointSampling(Function):
    """xyz (B, N, 3) -> (B, npoint) int32"""

    @staticmethod
    def forward(ctx, xyz, npoint):
        C.require_contiguous(xyz)
        n_batch, n_pts = xyz.shape[:2]
        picked = C.empty_i32((n_batch, npoint), xyz.device)
        running = torch.full((n_batch, n_pts), 1e10, dtype=torch.float32, device=xyz.device)
        pointnet2.furthest_point_sampling_wrapper(n_batch, n_pts, npoint, xyz, running, picked)
        return picked

    @staticmethod
    def backward(ctx, *unused):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    """features (B, C, N) picked at idx (B, npoint) -> (B, C, npoint); differentiable in the features"""

    @staticmethod
    def forward(ctx, features, idx):
        C.require_contiguous(features, idx)
        n_batch, width, n_src = features.shape
        n_pick = idx.shape[1]
        out = C.empty_f32((n_batch, width, n_pick), features.device)
        pointnet2.gather_points_wrapper(n_batch, width, n_src, n_pick, features, idx, out)
        ctx.gather_state = (idx, width, n_src)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, width, n_src = ctx.gather_state
        n_batch, n_pick = idx.shape
        grad_features = C.zeros_f32((n_batch, width, n_src), grad_out.device)
        pointnet2.gather_points_grad_wrapper(n_batch, width, n_src, n_pick, grad_out.detach().contiguous(), idx, grad_features)
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """unknown (B, n, 3), known (B, m, 3) -> (distances (B, n, 3), indices into known (B, n, 3) int32)"""

    @staticmethod
    def forward(ctx, unknown, known):
        C.require_contiguous(unknown, known)
        n_batch, n_unknown = unknown.shape[:2]
        d2 = C.empty_f32((n_batch, n_unknown, 3), unknown.device)
        idx = C.empty_i32((n_batch, n_unknown, 3), unknown.device)
        pointnet2.three_nn_wrapper(n_batch, n_unknown, known.shape[1], unknown, known, d2, idx)
        return d2.sqrt(), idx

    @staticmethod
    def backward(ctx, *unused):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """features (B, c, m) blended through idx / weight (B, n, 3) -> (B, c, n); differentiable in the features"""

    @staticmethod
    def forward(ctx, features, idx, weight):
        C.require_contiguous(features, idx, weight)
        n_batch, width, n_known = features.shape
        n_out = idx.shape[1]
        ctx.interp_state = (idx, weight, n_known)
        out = C.empty_f32((n_batch, width, n_out), features.device)
        pointnet2.three_interpolate_wrapper(n_batch, width, n_known, n_out, features, idx, weight, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, n_known = ctx.interp_state
        n_batch, width, n_out = grad_out.shape
        grad_features = C.zeros_f32((n_batch, width, n_known), grad_out.device)
        pointnet2.three_interpolate_grad_wrapper(n_batch, width, n_out, n_known, grad_out.detach().contiguous(), idx, weight,
                                                 grad_features)
        return grad_features, None, None


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    """features (B, C, N) gathered through idx (B, npoint, nsample) -> (B, C, npoint, nsample); differentiable"""

    @staticmethod
    def forward(ctx, features, idx):
        C.require_contiguous(features, idx)
        n_batch, n_centre, nsample = idx.shape
        width, n_src = features.shape[1:]
        out = C.empty_f32((n_batch, width, n_centre, nsample), features.device)
        pointnet2.group_points_wrapper(n_batch, width, n_src, n_centre, nsample, features, idx, out)
        ctx.group_state = (idx, n_src)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, n_src = ctx.group_state
        n_batch, width, n_centre, nsample = grad_out.shape
        grad_features = C.zeros_f32((n_batch, width, n_src), grad_out.device)
        pointnet2.group_points_grad_wrapper(n_batch, width, n_src, n_centre, nsample, grad_out.detach().contiguous(), idx,
                                            grad_features)
        return grad_features, None


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    """(B, npoint, nsample) int32 neighbour indices; a ball without neighbours keeps the zero fill (no marker in this variant)"""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        C.require_contiguous(new_xyz, xyz)
        n_batch, n_src = xyz.shape[:2]
        n_centre = new_xyz.shape[1]
        idx = C.zeros_i32((n_batch, n_centre, nsample), xyz.device)
        pointnet2.ball_query_wrapper(n_batch, n_src, n_centre, radius, nsample, new_xyz, xyz, idx)
        return idx

    @staticmethod
    def backward(ctx, *unused):
        return (None,) * 4


ball_query = BallQuery.apply


def _with_coordinates(offsets, gathered, use_xyz):
    if gathered is None:
        return offsets
    return torch.cat((offsets, gathered), dim=1) if use_xyz else gathered


class QueryAndGroup(nn.Module):
    """ball query around new_xyz, neighbours' offsets (and features) -> (B, 3 [+ C], npoint, nsample)"""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        offsets = grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None and not self.use_xyz:
            raise AssertionError('nothing to group: no features and use_xyz=False')
        return _with_coordinates(offsets, None if features is None else grouping_operation(features, idx), self.use_xyz)


class GroupAll(nn.Module):
    """one group holding every point: (B, 3 [+ C], 1, N)"""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        coords = xyz.transpose(1, 2).unsqueeze(2)
        return _with_coordinates(coords, None if features is None else features.unsqueeze(2), self.use_xyz)
