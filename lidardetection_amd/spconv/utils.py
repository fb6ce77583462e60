"""spconv.utils.VoxelGenerator / VoxelGeneratorV2 (call site pcdet/datasets/processor/data_processor.py:48-80).

The reference runs this on the CPU inside DataLoader workers, one frame at a time; here `generate` runs the batched HIP
voxeliser on one frame (H2D, kernels, D2H).  For throughput, feed raw points to BatchVoxelizer on the device instead
(lidardetection_amd.voxelizer) — INTEGRATION.md."""
import numpy as np
import torch

from ..voxelizer import BatchVoxelizer, grid_size_of


class VoxelGeneratorV2(object):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, full_mean=False,
                 block_filtering=False, block_factor=8, block_size=3, height_threshold=0.1, height_high_threshold=2.0):
        assert not full_mean and not block_filtering, "options unused by the reference"
        self._voxel_size = np.array(voxel_size, dtype=np.float32)
        self._point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        self._max_num_points, self._max_voxels = max_num_points, max_voxels
        self._grid_size = grid_size_of(voxel_size, point_cloud_range)
        self._vz = {}

    def _voxelizer(self, c):
        if c not in self._vz:
            self._vz[c] = BatchVoxelizer(self._voxel_size, self._point_cloud_range, self._max_num_points, self._max_voxels, c)
        return self._vz[c]

    def generate(self, points, max_voxels=None):
        assert max_voxels is None or max_voxels == self._max_voxels
        pts = np.ascontiguousarray(points, dtype=np.float32)
        out = self._voxelizer(pts.shape[1]).voxelize_frames([pts])
        coords = out["voxel_coords"][:, 1:4].cpu().numpy()           # (z, y, x); the batch column is added by collate_batch
        num = out["voxel_num_points"].cpu().numpy()
        voxels = out["voxels"].cpu().numpy()
        P = self._max_num_points
        mask = (np.arange(P, dtype=np.int32)[None, :] < num[:, None])
        return {"voxels": voxels, "coordinates": coords, "num_points_per_voxel": num,
                "voxel_point_mask": mask.reshape(-1, P, 1).astype(voxels.dtype), "voxel_num": len(voxels)}

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
