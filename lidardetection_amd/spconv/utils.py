"""spconv.utils.VoxelGenerator / VoxelGeneratorV2 (call site pcdet/datasets/processor/data_processor.py:48-80).

The reference calls `generate(points)` with a numpy array inside forked DataLoader workers (pcdet/datasets/__init__.py:73,
tools/train.py:27: 8 workers by default), one frame at a time, and gets numpy arrays back.  A forked child must not touch the
GPU, so that call runs the HOST generator of this library (`lidar_voxelize_cpu`, csrc/cpu_ops.hip: hash map over the occupied
cells, no dense coordinate grid, no HIP call).  A CUDA tensor is voxelised on the device instead (`lidar_voxelize`, one frame)
and comes back as CUDA tensors.  Both give the sequential scan's result bit for bit (tests).  For throughput, skip the worker
side altogether and feed raw points to BatchVoxelizer on the device (lidardetection_amd.voxelizer) — INTEGRATION.md."""
import ctypes as C

import numpy as np

from .. import _lib
from ..voxelizer import BatchVoxelizer, grid_size_of


class VoxelGeneratorV2(object):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, full_mean=False,
                 block_filtering=False, block_factor=8, block_size=3, height_threshold=0.1, height_high_threshold=2.0):
        assert not full_mean and not block_filtering, "options unused by the reference"
        self._voxel_size = np.array(voxel_size, dtype=np.float32)
        self._point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        self._max_num_points, self._max_voxels = int(max_num_points), int(max_voxels)
        self._grid_size = grid_size_of(voxel_size, point_cloud_range)
        self._grid_i32 = np.ascontiguousarray(self._grid_size, dtype=np.int32)
        self._vz = {}

    def _voxelizer(self, c):
        if c not in self._vz:
            self._vz[c] = BatchVoxelizer(self._voxel_size, self._point_cloud_range, self._max_num_points, self._max_voxels, c)
        return self._vz[c]

    def _generate_host(self, points, max_voxels):
        pts = np.ascontiguousarray(points, dtype=np.float32)
        if pts.ndim != 2 or pts.shape[1] < 3:
            raise _lib.LidarHipError("generate: points must be (N, C >= 3)")
        n, c = pts.shape
        P = self._max_num_points
        L = _lib.lib()
        voxels = np.empty((max_voxels, P, c), np.float32)             # only the produced rows are ever written / returned
        coords = np.empty((max_voxels, 3), np.int32)
        num = np.empty((max_voxels,), np.int32)
        sb = int(L.lidar_voxelize_cpu_scratch_bytes(n))
        scratch = np.empty((sb,), np.uint8)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        nv = L.lidar_voxelize_cpu(vp(pts), n, c, vp(self._point_cloud_range), vp(self._voxel_size), vp(self._grid_i32), P,
                                  max_voxels, vp(voxels), vp(coords), vp(num), vp(scratch), sb)
        if nv < 0:
            raise _lib.LidarHipError(f"lidar_voxelize_cpu failed with status {nv}")
        return voxels[:nv], coords[:nv], num[:nv]

    def _generate_device(self, points, max_voxels):
        if max_voxels != self._max_voxels:
            raise _lib.LidarHipError("generate(max_voxels=...) on a CUDA tensor must equal the constructor's max_voxels")
        out = self._voxelizer(points.shape[1]).voxelize_frames([points], device=points.device)
        return out["voxels"], out["voxel_coords"][:, 1:4].contiguous(), out["voxel_num_points"]   # the batch column is collate's

    def generate(self, points, max_voxels=None):
        """-> dict(voxels (V, P, C), coordinates (V, 3) [z, y, x], num_points_per_voxel (V), voxel_point_mask (V, P, 1), voxel_num)"""
        mv = self._max_voxels if max_voxels is None else int(max_voxels)
        P = self._max_num_points
        if hasattr(points, "is_cuda") and points.is_cuda:
            import torch
            voxels, coords, num = self._generate_device(points.float().contiguous(), mv)
            mask = (torch.arange(P, dtype=torch.int32, device=num.device)[None, :] < num[:, None]).view(-1, P, 1).to(voxels.dtype)
        else:
            voxels, coords, num = self._generate_host(points.numpy() if hasattr(points, "numpy") else points, mv)
            mask = (np.arange(P, dtype=np.int32)[None, :] < num[:, None]).reshape(-1, P, 1).astype(voxels.dtype)
        return {"voxels": voxels, "coordinates": coords, "num_points_per_voxel": num, "voxel_point_mask": mask,
                "voxel_num": int(voxels.shape[0])}

    @property
    def voxel_size(self):
        return self._voxel_size

    @property
    def max_num_points_per_voxel(self):
        return self._max_num_points

    @property
    def point_cloud_range(self):
        return self._point_cloud_range

    @property
    def grid_size(self):
        return self._grid_size


class VoxelGenerator(VoxelGeneratorV2):
    """v1 interface: generate() returns the 3-tuple (voxels, coordinates, num_points_per_voxel)."""

    def generate(self, points, max_voxels=None):
        r = super().generate(points, max_voxels)
        return r["voxels"], r["coordinates"], r["num_points_per_voxel"]
