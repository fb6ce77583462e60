"""`pointnet2_stack_cuda` — same entry points as pcdet/ops/pointnet2/pointnet2_stack/src/pointnet2_api.cpp:10-21."""
from .. import _lib

_S = _lib.stream
_p = _lib.ptr
GRID_MIN_POINTS = 512          # candidates per batch element from which ball queries go through the cell grid


def ball_query_wrapper(B, M, radius, nsample, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx):
    """ball_query_wrapper_stack (ball_query.cpp:31-47): fills idx (M, nsample) int32 (zero-filled by the caller)."""
    _lib.require_cuda(new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx)
    _lib.require_last(new_xyz, 3, "new_xyz")
    _lib.require_last(xyz, 3, "xyz")
    if xyz.shape[0] >= GRID_MIN_POINTS * B and nsample <= 64 and radius > 0:
        # enough candidates per batch element to pay for a binning pass: same lists through the cell grid
        return ball_query_grid_wrapper(B, M, radius, nsample, None, None, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx, None)
    _lib.check(_lib.lib().lidar_ball_query_stack(B, M, float(radius), nsample, _p(new_xyz), _p(new_xyz_batch_cnt), _p(xyz),
                                                 _p(xyz_batch_cnt), _p(idx), _S()), "lidar_ball_query_stack")
    return 1


def ball_query2_wrapper(B, M, radius_a, nsample_a, radius_b, nsample_b, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_a, idx_b):
    """not in the reference's module: two radii in one pass (include/lidar_hip.h: lidar_ball_query_stack2)"""
    _lib.require_cuda(new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_a, idx_b)
    _lib.check(_lib.lib().lidar_ball_query_stack2(B, M, float(radius_a), nsample_a, float(radius_b), nsample_b, _p(new_xyz),
                                                  _p(new_xyz_batch_cnt), _p(xyz), _p(xyz_batch_cnt), _p(idx_a), _p(idx_b), _S()),
               "lidar_ball_query_stack2")
    return 1


def ball_query_grid_wrapper(B, M, radius_a, nsample_a, radius_b, nsample_b, new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_a, idx_b):
    """not in the reference's module: the same lists as ball_query_wrapper / ball_query2_wrapper (idx_b None: one radius) through
    a cell grid over the candidates (include/lidar_hip.h: lidar_ball_query_stack_grid)"""
    from .. import workspace
    _lib.require_cuda(new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idx_a, idx_b)
    L = _lib.lib()
    N = xyz.shape[0]
    nbytes = L.lidar_ball_query_grid_workspace_bytes(B, N)
    ws = workspace.get("ball_query_grid", nbytes, xyz.device)
    _lib.check(L.lidar_ball_query_stack_grid(B, M, N, float(radius_a), nsample_a, float(radius_b or 0.0), nsample_b or 0, _p(new_xyz),
                                             _p(new_xyz_batch_cnt), _p(xyz), _p(xyz_batch_cnt), _p(idx_a), _p(idx_b), _p(ws), nbytes,
                                             _S()), "lidar_ball_query_stack_grid")
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    """sampling.cpp: points (b,n,3), temp (b,n) == 1e10, idx (b,m) int32."""
    _lib.require_cuda(points, temp, idx)
    _lib.require_last(points, 3, "points")
    _lib.check(_lib.lib().lidar_furthest_point_sampling(b, n, m, _p(points), _p(temp), _p(idx), _S()), "lidar_furthest_point_sampling")
    return 1


def group_points_wrapper(B, M, C, nsample, features, features_batch_cnt, idx, idx_batch_cnt, out):
    _lib.require_cuda(features, features_batch_cnt, idx, idx_batch_cnt, out)
    _lib.check(_lib.lib().lidar_group_points_stack(B, M, C, nsample, _p(features), _p(features_batch_cnt), _p(idx),
                                                   _p(idx_batch_cnt), _p(out), _S()), "lidar_group_points_stack")
    return 1


def group_rows_wrapper(B, M, C, nsample, use_xyz, stride, xyz, new_xyz, features, features_batch_cnt, idx, idx_batch_cnt, out):
    """not in the reference's module: (M, nsample, stride) row-major groups for the inference MLP (include/lidar_hip.h)"""
    _lib.require_cuda(xyz, new_xyz, features, features_batch_cnt, idx, idx_batch_cnt, out)
    _lib.check(_lib.lib().lidar_group_rows_stack(B, M, C, nsample, int(bool(use_xyz)), stride, _p(xyz), _p(new_xyz), _p(features),
                                                 _p(features_batch_cnt), _p(idx), _p(idx_batch_cnt), _p(out), _S()),
               "lidar_group_rows_stack")
    return 1


def group_rows_affine_wrapper(B, M, H, nsample, table, query_term, empty_row, features_batch_cnt, idx, idx_batch_cnt, out):
    """not in the reference's module: relu(table[idx] - query_term[m]) rows (include/lidar_hip.h: lidar_group_rows_affine_stack)"""
    _lib.require_cuda(table, query_term, empty_row, features_batch_cnt, idx, idx_batch_cnt, out)
    _lib.check(_lib.lib().lidar_group_rows_affine_stack(B, M, H, nsample, _p(table), _p(query_term), _p(empty_row),
                                                        _p(features_batch_cnt), _p(idx), _p(idx_batch_cnt), _p(out), _S()),
               "lidar_group_rows_affine_stack")
    return 1


def sa_layer2_max_supported(H1, H2, nsample):
    return bool(_lib.lib().lidar_sa_layer2_max_supported(int(H1), int(H2), int(nsample)))


def sa_layer2_max_wrapper(B, M, nsample, table, query_term, empty_row, w2, b2, features_batch_cnt, idx, idx_batch_cnt, out):
    """not in the reference's module: layer-1 gather + second layer (MFMA) + max over the samples in one kernel
    (include/lidar_hip.h: lidar_sa_layer2_max_stack)"""
    _lib.require_cuda(table, query_term, empty_row, w2, b2, features_batch_cnt, idx, idx_batch_cnt, out)
    _lib.check(_lib.lib().lidar_sa_layer2_max_stack(B, M, table.shape[1], w2.shape[1], nsample, _p(table), _p(query_term), _p(empty_row),
                                                    _p(w2), _p(b2), _p(features_batch_cnt), _p(idx), _p(idx_batch_cnt), _p(out), _S()),
               "lidar_sa_layer2_max_stack")
    return 1


def group_points_grad_wrapper(B, M, C, N, nsample, grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features):
    _lib.require_cuda(grad_out, idx, idx_batch_cnt, features_batch_cnt, grad_features)
    _lib.check(_lib.lib().lidar_group_points_grad_stack(B, M, C, N, nsample, _p(grad_out), _p(idx), _p(idx_batch_cnt),
                                                        _p(features_batch_cnt), _p(grad_features), _S()),
               "lidar_group_points_grad_stack")
    return 1


def three_nn_wrapper(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx):
    _lib.require_cuda(unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx)
    _lib.require_last(unknown, 3, "unknown")
    _lib.require_last(known, 3, "known")
    _lib.check(_lib.lib().lidar_three_nn_stack(unknown_batch_cnt.shape[0], unknown.shape[0], _p(unknown), _p(unknown_batch_cnt),
                                               _p(known), _p(known_batch_cnt), _p(dist2), _p(idx), _S()), "lidar_three_nn_stack")


def three_interpolate_wrapper(features, idx, weight, out):
    _lib.require_cuda(features, idx, weight, out)
    _lib.check(_lib.lib().lidar_three_interpolate_stack(idx.shape[0], features.shape[1], _p(features), _p(idx), _p(weight), _p(out),
                                                        _S()), "lidar_three_interpolate_stack")


def three_interpolate_grad_wrapper(grad_out, idx, weight, grad_features):
    _lib.require_cuda(grad_out, idx, weight, grad_features)
    _lib.check(_lib.lib().lidar_three_interpolate_grad_stack(idx.shape[0], grad_out.shape[1], _p(grad_out), _p(idx), _p(weight),
                                                             _p(grad_features), _S()), "lidar_three_interpolate_grad_stack")
