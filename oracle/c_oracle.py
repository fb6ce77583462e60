"""ctypes front-end of oracle/src/*.c.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        so = _build.SO
        if os.path.isdir(os.path.join(_build.HERE, "src")):
            so = _build.build()
        _lib = C.CDLL(so)
        _lib.orc_box_overlap.restype = C.c_float
        _lib.orc_iou_bev.restype = C.c_float
        _lib.orc_iou_normal.restype = C.c_float
        for f in ("orc_nms_greedy", "orc_nms", "orc_voxelize"):
            getattr(_lib, f).restype = C.c_int
    return _lib


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- iou3d_nms
def pairwise(boxes_a, boxes_b, mode):
    """mode 0: BEV overlap area, 1: rotated BEV IoU, 2: axis-aligned BEV IoU -> (N, M) f32."""
    a, b = _f32(boxes_a), _f32(boxes_b)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().orc_pairwise(_p(a), C.c_int(a.shape[0]), _p(b), C.c_int(b.shape[0]), C.c_int(mode), _p(out))
    return out


def nms_mask(boxes_sorted, thresh, normal=False):
    b = _f32(boxes_sorted)
    n = b.shape[0]
    cb = (n + 63) // 64
    mask = np.zeros((n, cb), np.uint64)
    lib().orc_nms_mask(_p(b), C.c_int(n), C.c_float(thresh), C.c_int(int(normal)), _p(mask))
    return mask


def nms_greedy(mask):
    n = mask.shape[0]
    keep = np.zeros((max(n, 1),), np.int64)
    k = lib().orc_nms_greedy(_p(np.ascontiguousarray(mask)), C.c_int(n), _p(keep))
    return keep[:k]


def nms_sorted(boxes_sorted, thresh, normal=False):
    """Reference native entry nms_gpu / nms_normal_gpu on already-sorted boxes -> keep positions."""
    b = _f32(boxes_sorted)
    n = b.shape[0]
    keep = np.zeros((max(n, 1),), np.int64)
    k = lib().orc_nms(_p(b), C.c_int(n), C.c_float(thresh), C.c_int(int(normal)), _p(keep))
    return keep[:k]


def nms(boxes, scores, thresh, pre_maxsize=None, normal=False):
    """Python-level wrapper semantics (iou3d_nms_utils.py:84-116): sort desc, cut, nms, map back."""
    order = np.argsort(-np.asarray(scores, dtype=np.float32), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = nms_sorted(np.asarray(boxes)[order], thresh, normal)
    return order[keep]


# ---------------------------------------------------------------- voxelise
_grid_cache = {}


def voxelize(points, voxel_size, pc_range, max_points, max_voxels):
    """spconv VoxelGeneratorV2.generate restatement -> voxels (V,P,C) f32, coords (V,3) i32 zyx, num (V,) i32."""
    pts = _f32(points)
    n, c = pts.shape
    rng = np.asarray(pc_range, np.float32)
    vs = np.asarray(voxel_size, np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int64)
    g32 = grid.astype(np.int32)
    key = tuple(int(x) for x in grid)
    if key not in _grid_cache:
        _grid_cache.clear()  # one persistent map at a time (SECOND's is 360 MB)
        _grid_cache[key] = np.full(int(np.prod(grid)), -1, np.int32)
    cmap = _grid_cache[key]
    voxels = np.zeros((max_voxels, max_points, c), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    num = np.zeros((max_voxels,), np.int32)
    v = lib().orc_voxelize(_p(pts), C.c_int(n), C.c_int(c), _p(rng), _p(vs), _p(g32), C.c_int(max_points),
                           C.c_int(max_voxels), _p(voxels), _p(coors), _p(num), _p(cmap))
    return voxels[:v], coors[:v], num[:v]


# ---------------------------------------------------------------- point-set operators (points_oracle.c)
def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ci(v):
    return C.c_int(int(v))


def points_in_boxes_cpu(boxes, pts):
    b, p = _f32(boxes), _f32(pts)
    out = np.zeros((len(b), len(p)), np.int32)
    lib().orc_points_in_boxes_cpu(_p(b), _ci(len(b)), _p(p), _ci(len(p)), _p(out))
    return out


def points_in_boxes_gpu(boxes, pts):
    """boxes (B,T,7), pts (B,P,3) -> (B,P) first containing box or -1."""
    b, p = _f32(boxes), _f32(pts)
    B, T, P = b.shape[0], b.shape[1], p.shape[1]
    out = np.zeros((B, P), np.int32)
    lib().orc_points_in_boxes_gpu(_p(b), _p(p), _ci(B), _ci(T), _ci(P), _p(out))
    return out


def roiaware_pool3d(rois, pts, feat, out_size, max_pts, pool_method):
    rois, pts, feat = _f32(rois), _f32(pts), _f32(feat)
    R, P, Cc = len(rois), len(pts), feat.shape[1]
    ox, oy, oz = out_size
    argmax = np.zeros((R, ox, oy, oz, Cc), np.int32)
    pidx = np.zeros((R, ox, oy, oz, max_pts), np.int32)
    pooled = np.zeros((R, ox, oy, oz, Cc), np.float32)
    lib().orc_roiaware_pool3d(_p(rois), _ci(R), _p(pts), _p(feat), _ci(P), _ci(Cc), _ci(ox), _ci(oy), _ci(oz), _ci(max_pts),
                              _ci(pool_method), _p(argmax), _p(pidx), _p(pooled))
    return pooled, argmax, pidx


def roiaware_pool3d_backward(pidx, argmax, grad_out, npts, pool_method):
    pidx, argmax, g = _i32(pidx), _i32(argmax), _f32(grad_out)
    R, ox, oy, oz, Cc = g.shape
    gi = np.zeros((npts, Cc), np.float32)
    lib().orc_roiaware_pool3d_backward(_p(pidx), _p(argmax), _p(g), _ci(R), _ci(ox), _ci(oy), _ci(oz), _ci(Cc),
                                       _ci(pidx.shape[-1]), _ci(pool_method), _p(gi))
    return gi


def roipoint_pool3d(xyz, boxes_enlarged, feat, S):
    xyz, bx, feat = _f32(xyz), _f32(boxes_enlarged), _f32(feat)
    B, N, M, Cc = xyz.shape[0], xyz.shape[1], bx.shape[1], feat.shape[2]
    pooled = np.zeros((B, M, S, 3 + Cc), np.float32)
    empty = np.zeros((B, M), np.int32)
    lib().orc_roipoint_pool3d(_p(xyz), _p(bx), _p(feat), _ci(B), _ci(N), _ci(M), _ci(Cc), _ci(S), _p(pooled), _p(empty))
    return pooled, empty


def ball_query_stack(radius, nsample, xyz, xyz_cnt, new_xyz, new_cnt):
    xyz, new_xyz, xc, nc = _f32(xyz), _f32(new_xyz), _i32(xyz_cnt), _i32(new_cnt)
    M = len(new_xyz)
    idx = np.zeros((M, nsample), np.int32)
    lib().orc_ball_query_stack(_ci(len(xc)), _ci(M), C.c_float(radius), _ci(nsample), _p(new_xyz), _p(nc), _p(xyz), _p(xc), _p(idx))
    return idx


def group_points_stack(feat, feat_cnt, idx, idx_cnt):
    feat, idx, fc, ic = _f32(feat), _i32(idx), _i32(feat_cnt), _i32(idx_cnt)
    M, ns, Cc = idx.shape[0], idx.shape[1], feat.shape[1]
    out = np.zeros((M, Cc, ns), np.float32)
    lib().orc_group_points_stack(_ci(len(fc)), _ci(M), _ci(Cc), _ci(ns), _p(feat), _p(fc), _p(idx), _p(ic), _p(out))
    return out


def group_points_grad_stack(grad_out, idx, idx_cnt, feat_cnt, N):
    g, idx, fc, ic = _f32(grad_out), _i32(idx), _i32(feat_cnt), _i32(idx_cnt)
    M, Cc, ns = g.shape
    gf = np.zeros((N, Cc), np.float32)
    lib().orc_group_points_grad_stack(_ci(len(fc)), _ci(M), _ci(Cc), _ci(ns), _p(g), _p(idx), _p(ic), _p(fc), _p(gf))
    return gf


def fps(xyz, m):
    """xyz (B,N,3) -> idx (B,m) int32 (temp initialised to 1e10 as the reference wrapper does)."""
    x = _f32(xyz)
    B, N = x.shape[0], x.shape[1]
    temp = np.full((B, N), 1e10, np.float32)
    idx = np.zeros((B, m), np.int32)
    lib().orc_fps(_ci(B), _ci(N), _ci(m), _p(x), _p(temp), _p(idx))
    return idx


def three_nn_stack(unknown, unk_cnt, known, known_cnt):
    u, k, uc, kc = _f32(unknown), _f32(known), _i32(unk_cnt), _i32(known_cnt)
    N = len(u)
    d2, idx = np.zeros((N, 3), np.float32), np.zeros((N, 3), np.int32)
    lib().orc_three_nn_stack(_ci(len(uc)), _ci(N), _p(u), _p(uc), _p(k), _p(kc), _p(d2), _p(idx))
    return d2, idx


def three_interpolate_stack(feat, idx, w):
    feat, idx, w = _f32(feat), _i32(idx), _f32(w)
    out = np.zeros((len(idx), feat.shape[1]), np.float32)
    lib().orc_three_interpolate_stack(_ci(len(idx)), _ci(feat.shape[1]), _p(feat), _p(idx), _p(w), _p(out))
    return out


def three_interpolate_grad_stack(grad_out, idx, w, M):
    g, idx, w = _f32(grad_out), _i32(idx), _f32(w)
    gf = np.zeros((M, g.shape[1]), np.float32)
    lib().orc_three_interpolate_grad_stack(_ci(len(idx)), _ci(g.shape[1]), _p(g), _p(idx), _p(w), _p(gf))
    return gf


def ball_query_batch(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, m = xyz.shape[0], xyz.shape[1], new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().orc_ball_query_batch(_ci(b), _ci(n), _ci(m), C.c_float(radius), _ci(nsample), _p(new_xyz), _p(xyz), _p(idx))
    return idx


def group_points_batch(points, idx):
    pts, idx = _f32(points), _i32(idx)
    b, c, n = pts.shape
    np_, ns = idx.shape[1], idx.shape[2]
    out = np.zeros((b, c, np_, ns), np.float32)
    lib().orc_group_points_batch(_ci(b), _ci(c), _ci(n), _ci(np_), _ci(ns), _p(pts), _p(idx), _p(out))
    return out


def group_points_grad_batch(grad_out, idx, n):
    g, idx = _f32(grad_out), _i32(idx)
    b, c, np_, ns = g.shape
    gp = np.zeros((b, c, n), np.float32)
    lib().orc_group_points_grad_batch(_ci(b), _ci(c), _ci(n), _ci(np_), _ci(ns), _p(g), _p(idx), _p(gp))
    return gp


def gather_points_batch(points, idx):
    pts, idx = _f32(points), _i32(idx)
    b, c, n = pts.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().orc_gather_points_batch(_ci(b), _ci(c), _ci(n), _ci(m), _p(pts), _p(idx), _p(out))
    return out


def gather_points_grad_batch(grad_out, idx, n):
    g, idx = _f32(grad_out), _i32(idx)
    b, c, m = g.shape
    gp = np.zeros((b, c, n), np.float32)
    lib().orc_gather_points_grad_batch(_ci(b), _ci(c), _ci(n), _ci(m), _p(g), _p(idx), _p(gp))
    return gp


def three_nn_batch(unknown, known):
    u, k = _f32(unknown), _f32(known)
    b, n, m = u.shape[0], u.shape[1], k.shape[1]
    d2, idx = np.zeros((b, n, 3), np.float32), np.zeros((b, n, 3), np.int32)
    lib().orc_three_nn_batch(_ci(b), _ci(n), _ci(m), _p(u), _p(k), _p(d2), _p(idx))
    return d2, idx


def three_interpolate_batch(points, idx, w):
    pts, idx, w = _f32(points), _i32(idx), _f32(w)
    b, c, m = pts.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().orc_three_interpolate_batch(_ci(b), _ci(c), _ci(m), _ci(n), _p(pts), _p(idx), _p(w), _p(out))
    return out


def three_interpolate_grad_batch(grad_out, idx, w, m):
    g, idx, w = _f32(grad_out), _i32(idx), _f32(w)
    b, c, n = g.shape
    gp = np.zeros((b, c, m), np.float32)
    lib().orc_three_interpolate_grad_batch(_ci(b), _ci(c), _ci(n), _ci(m), _p(g), _p(idx), _p(w), _p(gp))
    return gp


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """KITTI-eval rotated IoU (rotate_iou.py:290-330), boxes (N,5) / query_boxes (K,5) [x, y, w, l, angle] -> (N,K) f32."""
    b, q = _f32(boxes), _f32(query_boxes)
    out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float32)
    if b.shape[0] and q.shape[0]:
        lib().orc_rotate_iou_eval(_p(b), _ci(b.shape[0]), _p(q), _ci(q.shape[0]), _ci(criterion), _p(out))
    return out
