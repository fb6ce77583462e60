"""`roipoint_pool3d_cuda` — same entry point as pcdet/ops/roipoint_pool3d/src/roipoint_pool3d.cpp:57-59."""
from .. import _lib


def forward(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag):
    """roipool3d_gpu (:23-54): xyz (B,N,3), boxes3d (B,M,7), pts_feature (B,N,C), pooled (B,M,S,3+C) zeros, flag (B,M) zeros."""
    _lib.require_cuda(xyz, boxes3d, pts_feature, pooled_features, pooled_empty_flag)
    _lib.require_last(xyz, 3, "xyz")
    _lib.require_last(boxes3d, 7, "boxes3d")
    B, N, M, C, S = xyz.shape[0], xyz.shape[1], boxes3d.shape[1], pts_feature.shape[2], pooled_features.shape[2]
    _lib.check(_lib.lib().lidar_roipoint_pool3d_forward(B, N, M, C, S, _lib.ptr(xyz), _lib.ptr(boxes3d), _lib.ptr(pts_feature),
                                                        _lib.ptr(pooled_features), _lib.ptr(pooled_empty_flag), _lib.stream()),
               "lidar_roipoint_pool3d_forward")
    return 1
