"""RoI-aware pooling and point-in-box queries under the reference's public names (pcdet/ops/roiaware_pool3d/
roiaware_pool3d_utils.py:9-107), bound to lidardetection_amd.ext.roiaware_pool3d_cuda.
Boxes: [x, y, z, dx, dy, dz, heading], (x, y, z) = box centre.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from ...utils import common_utils
from ....ext import roiaware_pool3d_cuda as _native

_POOL_CODES = {'max': 0, 'avg': 1}


def points_in_boxes_cpu(points, boxes):
    """host data (tensor or numpy): points (P, 3), boxes (N, 7) -> (N, P) 0/1 membership; faces count as inside up to the
    reference's 1e-2 tolerance (reference :9-25)"""
    if boxes.shape[1] != 7 or points.shape[1] != 3:
        raise AssertionError('expected boxes (N, 7) and points (P, 3)')
    pts_t, as_numpy = common_utils.check_numpy_to_torch(points)
    box_t, _ = common_utils.check_numpy_to_torch(boxes)
    member = torch.zeros((box_t.shape[0], pts_t.shape[0]), dtype=torch.int32)
    _native.points_in_boxes_cpu(box_t.float().contiguous(), pts_t.float().contiguous(), member)
    return member.numpy() if as_numpy else member


def points_in_boxes_gpu(points, boxes):
    """points (B, M, 3), boxes (B, T, 7) on the device -> (B, M) int32: lowest index of a box holding the point, else -1
    (reference :28-41)"""
    if points.shape[0] != boxes.shape[0] or boxes.shape[2] != 7 or points.shape[2] != 3:
        raise AssertionError('expected points (B, M, 3) and boxes (B, T, 7) with the same B')
    owner = torch.full(points.shape[:2], -1, dtype=torch.int32, device=points.device)
    _native.points_in_boxes_gpu(boxes.contiguous(), points.contiguous(), owner)
    return owner


def _grid3(out_size):
    if isinstance(out_size, int):
        return out_size, out_size, out_size
    if len(out_size) != 3 or not all(isinstance(v, int) for v in out_size):
        raise AssertionError('out_size must be an int or three ints')
    return tuple(out_size)


class RoIAwarePool3dFunction(Function):
    """rois (N, 7), pts (P, 3), pts_feature (P, C) -> pooled (N, gx, gy, gz, C); differentiable in pts_feature
    (reference :55-107)."""

    @staticmethod
    def forward(ctx, rois, pts, pts_feature, out_size, max_pts_each_voxel, pool_method):
        if rois.shape[1] != 7 or pts.shape[1] != 3:
            raise AssertionError('expected rois (N, 7) and pts (P, 3)')
        gx, gy, gz = _grid3(out_size)
        n_roi, n_ch = rois.shape[0], pts_feature.shape[-1]
        cell = (n_roi, gx, gy, gz)
        pooled = pts_feature.new_zeros(cell + (n_ch,))
        winner = torch.zeros(cell + (n_ch,), dtype=torch.int32, device=pts_feature.device)
        members = torch.zeros(cell + (max_pts_each_voxel,), dtype=torch.int32, device=pts_feature.device)
        code = _POOL_CODES[pool_method]
        _native.forward(rois.contiguous(), pts.contiguous(), pts_feature.contiguous(), winner, members, pooled, code)
        ctx.roiaware_pool3d_for_backward = (members, winner, code, pts.shape[0], n_ch)   # the reference's attribute name
        return pooled

    @staticmethod
    def backward(ctx, grad_out):
        members, winner, code, n_pts, n_ch = ctx.roiaware_pool3d_for_backward
        grad_feat = grad_out.new_zeros((n_pts, n_ch))
        _native.backward(members, winner, grad_out.contiguous(), grad_feat, code)
        return None, None, grad_feat, None, None, None


class RoIAwarePool3d(nn.Module):
    def __init__(self, out_size, max_pts_each_voxel=128):
        super().__init__()
        self.out_size, self.max_pts_each_voxel = out_size, max_pts_each_voxel

    def forward(self, rois, pts, pts_feature, pool_method='max'):
        if pool_method not in _POOL_CODES:
            raise AssertionError("pool_method must be 'max' or 'avg'")
        return RoIAwarePool3dFunction.apply(rois, pts, pts_feature, self.out_size, self.max_pts_each_voxel, pool_method)
