"""`pointnet2_batch_cuda` — same entry points as pcdet/ops/pointnet2/pointnet2_batch/src/pointnet2_api.cpp:10-24."""
from .. import _lib

_S = _lib.stream
_p = _lib.ptr


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    _lib.require_cuda(new_xyz, xyz, idx)
    _lib.check(_lib.lib().lidar_ball_query_batch(b, n, m, float(radius), nsample, _p(new_xyz), _p(xyz), _p(idx), _S()),
               "lidar_ball_query_batch")
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _lib.require_cuda(points, idx, out)
    _lib.check(_lib.lib().lidar_group_points_batch(b, c, n, npoints, nsample, _p(points), _p(idx), _p(out), _S()),
               "lidar_group_points_batch")
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _lib.require_cuda(grad_out, idx, grad_points)
    _lib.check(_lib.lib().lidar_group_points_grad_batch(b, c, n, npoints, nsample, _p(grad_out), _p(idx), _p(grad_points), _S()),
               "lidar_group_points_grad_batch")
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _lib.require_cuda(points, idx, out)
    _lib.check(_lib.lib().lidar_gather_points_batch(b, c, n, npoints, _p(points), _p(idx), _p(out), _S()), "lidar_gather_points_batch")
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _lib.require_cuda(grad_out, idx, grad_points)
    _lib.check(_lib.lib().lidar_gather_points_grad_batch(b, c, n, npoints, _p(grad_out), _p(idx), _p(grad_points), _S()),
               "lidar_gather_points_grad_batch")
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _lib.require_cuda(points, temp, idx)
    _lib.check(_lib.lib().lidar_furthest_point_sampling(b, n, m, _p(points), _p(temp), _p(idx), _S()), "lidar_furthest_point_sampling")
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _lib.require_cuda(unknown, known, dist2, idx)
    _lib.check(_lib.lib().lidar_three_nn_batch(b, n, m, _p(unknown), _p(known), _p(dist2), _p(idx), _S()), "lidar_three_nn_batch")


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _lib.require_cuda(points, idx, weight, out)
    _lib.check(_lib.lib().lidar_three_interpolate_batch(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out), _S()),
               "lidar_three_interpolate_batch")


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _lib.require_cuda(grad_out, idx, weight, grad_points)
    _lib.check(_lib.lib().lidar_three_interpolate_grad_batch(b, c, n, m, _p(grad_out), _p(idx), _p(weight), _p(grad_points), _S()),
               "lidar_three_interpolate_grad_batch")
