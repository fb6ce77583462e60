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
