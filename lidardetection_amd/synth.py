"""Seeded synthetic inputs for the hot path.  numpy only; no GPU, no oracle.

The reference ships no data (KITTI / NuScenes must be downloaded), so every test and bench line in
this repo runs on these generators.  Shapes follow tools/cfgs/kitti_models/pointpillar.yaml:5,17-22.
"""
import numpy as np

PP_RANGE = [0.0, -39.68, -3.0, 69.12, 39.68, 1.0]
PP_VOXEL = [0.16, 0.16, 4.0]
SEC_RANGE = [0.0, -40.0, -3.0, 70.4, 40.0, 1.0]
SEC_VOXEL = [0.05, 0.05, 0.1]
NUS_RANGE = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
NUS_VOXEL = [0.1, 0.1, 0.2]

# class anchors (dx, dy, dz, z-centre): pointpillar.yaml:84,94,104 (Car, Pedestrian, Cyclist)
_ANCHORS = np.array([[3.9, 1.6, 1.56, -1.78 + 0.78], [0.8, 0.6, 1.73, -0.6 + 0.865],
                     [1.76, 0.6, 1.73, -0.6 + 0.865]], np.float32)


def cloud_uniform(seed=1000, n=20000, pc_range=PP_RANGE, c=4):
    """Uniform cloud: ~19k distinct PointPillar pillars => exercises the max_voxels=16000 cap."""
    r = np.random.default_rng(seed)
    lo, hi = np.asarray(pc_range[:3]), np.asarray(pc_range[3:])
    pts = np.empty((n, c), np.float32)
    for j in range(3):
        pts[:, j] = r.uniform(lo[j], hi[j], n).astype(np.float32)
    for j in range(3, c):
        pts[:, j] = r.uniform(0.0, 1.0, n).astype(np.float32)
    return pts


def cloud_ring(seed=2000, beams=64, az=312, c=4):
    """64-beam spinning-lidar-like cloud (19 968 pts): KITTI-like clustering, multi-point pillars."""
    r = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-24.8, 2.0, beams))
    azim = np.deg2rad(np.linspace(-40.5, 40.5, az))
    e, a = np.meshgrid(elev, azim, indexing="ij")
    e, a = e.ravel(), a.ravel()
    rng_ground = np.where(e < 0, np.minimum(1.73 / np.tan(np.maximum(-e, 1e-6)), 70.0), 0.0)
    rng_free = r.uniform(5.0, 70.0, e.shape[0])
    rr = np.where(e < 0, rng_ground, rng_free) * (1.0 + 0.01 * r.standard_normal(e.shape[0]))
    pts = np.empty((e.shape[0], c), np.float32)
    pts[:, 0] = (rr * np.cos(e) * np.cos(a)).astype(np.float32)
    pts[:, 1] = (rr * np.cos(e) * np.sin(a)).astype(np.float32)
    pts[:, 2] = (rr * np.sin(e)).astype(np.float32)
    for j in range(3, c):
        pts[:, j] = r.uniform(0.0, 1.0, e.shape[0]).astype(np.float32)
    return pts


def cloud_nus(seed=4000, n=30000):
    r = np.random.default_rng(seed)
    pts = np.empty((n, 5), np.float32)
    pts[:, 0] = r.uniform(-51.2, 51.2, n)
    pts[:, 1] = r.uniform(-51.2, 51.2, n)
    pts[:, 2] = r.uniform(-5.0, 3.0, n)
    pts[:, 3] = r.uniform(0.0, 1.0, n)
    pts[:, 4] = r.uniform(0.0, 0.5, n)
    return pts


def boxes_nms(seed=3000, objects=512, copies=8, pc_range=PP_RANGE):
    """512 objects x 8 jittered copies = 4096 boxes with all-distinct scores (sort-order independent)."""
    r = np.random.default_rng(seed)
    cls = r.integers(0, 3, objects)
    size = _ANCHORS[cls, :3] * r.uniform(0.9, 1.1, (objects, 3)).astype(np.float32)
    ctr = np.stack([r.uniform(pc_range[0], pc_range[3], objects), r.uniform(pc_range[1], pc_range[4], objects),
                    _ANCHORS[cls, 3]], 1)
    head = r.uniform(-np.pi, np.pi, objects)
    base = np.concatenate([ctr, size, head[:, None]], 1)
    boxes = np.repeat(base, copies, axis=0)
    boxes[:, 0:2] += 0.2 * r.standard_normal((objects * copies, 2))
    boxes[:, 6] += 0.05 * r.standard_normal(objects * copies)
    n = objects * copies
    scores = r.permutation(np.linspace(0.1, 0.99, n))
    return boxes.astype(np.float32), scores.astype(np.float32)


def boxes_random(seed, n, extent=20.0):
    r = np.random.default_rng(seed)
    return np.concatenate([r.uniform(0, extent, (n, 2)), r.uniform(-1, 1, (n, 1)), r.uniform(0.5, 5, (n, 2)),
                           r.uniform(1, 2, (n, 1)), r.uniform(-np.pi, np.pi, (n, 1))], 1).astype(np.float32)
