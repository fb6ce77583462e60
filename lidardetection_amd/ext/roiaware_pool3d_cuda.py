"""`roiaware_pool3d_cuda` — same entry points as pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp:172-177."""
import torch as _torch

from .. import _lib

_S = _lib.stream
_p = _lib.ptr


def forward(rois, pts, pts_feature, argmax, pts_idx_of_voxels, pooled_features, pool_method):
    """roiaware_pool3d_gpu (:29-66): outputs zero-initialised by the caller; pool_method 0 = max, 1 = avg."""
    _lib.require_cuda(rois, pts, pts_feature, argmax, pts_idx_of_voxels, pooled_features)
    _lib.require_last(rois, 7, "rois")
    _lib.require_last(pts, 3, "pts")
    R, P, C = rois.shape[0], pts.shape[0], pts_feature.shape[1]
    ox, oy, oz, K = pts_idx_of_voxels.shape[1:5]
    assert ox < 256 and oy < 256 and oz < 256  # (:53)
    _lib.check(_lib.lib().lidar_roiaware_pool3d_forward(R, P, C, K, ox, oy, oz, _p(rois), _p(pts), _p(pts_feature), _p(argmax),
                                                        _p(pts_idx_of_voxels), _p(pooled_features), int(pool_method), _S()),
               "lidar_roiaware_pool3d_forward")
    return 1


def backward(pts_idx_of_voxels, argmax, grad_out, grad_in, pool_method):
    """roiaware_pool3d_gpu_backward (:68-96): grad_in (P, C) zero-initialised by the caller."""
    _lib.require_cuda(pts_idx_of_voxels, argmax, grad_out, grad_in)
    R, ox, oy, oz, K = pts_idx_of_voxels.shape
    C = grad_out.shape[4]
    _lib.check(_lib.lib().lidar_roiaware_pool3d_backward(R, ox, oy, oz, C, K, _p(pts_idx_of_voxels), _p(argmax), _p(grad_out),
                                                         _p(grad_in), int(pool_method), _S()), "lidar_roiaware_pool3d_backward")
    return 1


def points_in_boxes_gpu(boxes, pts, box_idx_of_points):
    """(:98-118): boxes (B,T,7), pts (B,P,3), box_idx_of_points (B,P) int32 pre-filled with -1."""
    _lib.require_cuda(boxes, pts, box_idx_of_points)
    _lib.require_last(boxes, 7, "boxes")
    _lib.require_last(pts, 3, "pts")
    if box_idx_of_points.dtype != _torch.int32 or boxes.dtype != _torch.float32 or pts.dtype != _torch.float32:
        raise _lib.LidarHipError("points_in_boxes_gpu: boxes / pts float32, box_idx_of_points int32")
    _lib.check(_lib.lib().lidar_points_in_boxes(boxes.shape[0], boxes.shape[1], pts.shape[1], _p(boxes), _p(pts),
                                                _p(box_idx_of_points), _S()), "lidar_points_in_boxes")
    return 1


def points_in_boxes_cpu(boxes, pts, pts_indices):
    """roiaware_pool3d.cpp:143-168 — CPU tensors: boxes (N,7), pts (P,3), pts_indices (N,P) int32 0/1."""
    if boxes.is_cuda or pts.is_cuda or pts_indices.is_cuda:
        raise _lib.LidarHipError("points_in_boxes_cpu takes CPU tensors")
    if not pts_indices.is_contiguous():
        raise _lib.LidarHipError("points_in_boxes_cpu: pts_indices must be contiguous (it is written in place)")
    if boxes.dtype != _torch.float32 or pts.dtype != _torch.float32 or pts_indices.dtype != _torch.int32:
        raise _lib.LidarHipError("points_in_boxes_cpu: boxes / pts float32, pts_indices int32")
    _lib.require_last(boxes, 7, "boxes")
    _lib.require_last(pts, 3, "pts")
    boxes_c, pts_c = boxes.contiguous(), pts.contiguous()      # bound to locals: they must outlive the call
    _lib.check(_lib.lib().lidar_points_in_boxes_cpu(_p(boxes_c), boxes_c.shape[0], _p(pts_c), pts_c.shape[0],
                                                    _p(pts_indices)), "lidar_points_in_boxes_cpu")
    return 1
