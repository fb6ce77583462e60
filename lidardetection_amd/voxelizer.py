"""Batched on-device voxeliser (host side of lidar_voxelize, include/lidar_hip.h).

Replaces, for a whole batch in one launch sequence, what the reference does per frame on the CPU
inside DataLoader workers: spconv's VoxelGeneratorV2.generate (pcdet/datasets/processor/
data_processor.py:48-80) followed by collate_batch's concatenation + batch-index column
(pcdet/datasets/dataset.py:153-185).  Output is identical to that sequential path.
"""
import ctypes

import numpy as np
import torch

from . import _lib


def grid_size_of(voxel_size, point_cloud_range):
    """grid = round((range[3:6] - range[0:3]) / voxel_size) in fp32 (data_processor.py:61-62)."""
    r = np.asarray(point_cloud_range, np.float32)
    v = np.asarray(voxel_size, np.float32)
    return np.round((r[3:6] - r[0:3]) / v).astype(np.int64)


class BatchVoxelizer:
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels, num_point_features=4, algo=0):
        self.voxel_size = [float(np.float32(v)) for v in voxel_size]
        self.point_cloud_range = [float(np.float32(v)) for v in point_cloud_range]
        self.grid_size = grid_size_of(voxel_size, point_cloud_range)  # nx, ny, nz
        self.max_num_points = int(max_num_points)
        self.max_voxels = int(max_voxels)
        self.C = int(num_point_features)
        self.algo = int(algo)  # 0 auto, 3 LDS-binned (2 launches), 2 global hash (include/lidar_hip.h)
        self._range_h = _lib.host_f32(self.point_cloud_range)
        self._vs_h = _lib.host_f32(self.voxel_size)
        self._grid_h = _lib.host_i32(self.grid_size)
        self._ws = {}

    def _workspace(self, batch, n_max, device):
        key = (str(device), batch, n_max)
        ent = self._ws.get(key)
        if ent is None:
            L = _lib.lib()
            nbytes = L.lidar_voxelize_workspace_bytes(batch, n_max, self.max_voxels)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            _lib.check(L.lidar_voxelize_workspace_init(_lib.ptr(ws), nbytes, batch, n_max, self.max_voxels,
                                                       _lib.stream()), "lidar_voxelize_workspace_init")
            # the kernels mirror their sticky error bits into this pinned host word (device-visible on ROCm): the hot path
            # polls it on the host at no cost — no copy, no synchronisation (lidar_voxelize_set_error_mirror)
            mirror = torch.zeros(1, dtype=torch.int32).pin_memory()
            _lib.check(L.lidar_voxelize_set_error_mirror(_lib.ptr(ws), nbytes, batch, n_max, self.max_voxels, _lib.ptr(mirror),
                                                         _lib.stream()), "lidar_voxelize_set_error_mirror")
            ent = (ws, nbytes, mirror)
            self._ws = {key: ent}  # keep one workspace alive
        return ent

    def poll_error(self):
        """Sync-free look at the mirrored sticky flag of the live workspace: raises if an EARLIER call overflowed an LDS hash
        bin (its output was then wrong).  __call__ polls on entry, so a bad batch is reported one call late at the latest;
        callers that need it at once use check_error_flag (one device read) or voxelize_frames (redoes the batch)."""
        for ws, nbytes, mirror in self._ws.values():
            flag = int(mirror[0])
            if flag != 0:
                mirror[0] = 0
                self._ws = {}                 # a fresh workspace (and a cleared device flag) for whatever comes next
                raise _lib.LidarHipError(f"lidar_voxelize: an earlier call overflowed an LDS hash bin (flag {flag}): its voxels are "
                                         "incomplete; use algo=2 (global hash) for such input")

    def alloc_outputs(self, batch, device):
        rows = batch * self.max_voxels
        return {
            "voxels": torch.empty((rows, self.max_num_points, self.C), dtype=torch.float32, device=device),
            "voxel_coords": torch.empty((rows, 4), dtype=torch.int32, device=device),
            "voxel_num_points": torch.empty((rows,), dtype=torch.int32, device=device),
            "voxel_offsets": torch.empty((batch + 1,), dtype=torch.int32, device=device),
        }

    def __call__(self, points, point_offsets, n_max, compact=True, out=None, resident=False, host_offsets=None, timer=None):
        """points (sum N, C) f32 cuda; point_offsets (B+1) int32 cuda; n_max >= max frame size (host int).
        Returns dict(voxels, voxel_coords [b,z,y,x], voxel_num_points, voxel_offsets); rows beyond
        voxel_offsets[-1] are unspecified.  No host synchronisation.
        resident=True (with a persistent `out` that nobody else writes to between calls): the zero padding of `out` is kept
        from call to call and only the previous call's occupied slots are re-zeroed (include/lidar_hip.h, algo 4); the result is
        bit-identical, and rows beyond voxel_offsets[-1] are then all zero.
        host_offsets: the same B+1 offsets as a host sequence / numpy array, when the caller has them (a collate function does):
        they are handed to the launches as kernel arguments (lidar_voxelize_hostoff); results identical.
        timer: a handle from lidar_timer_create — this call's launches record their start / end into it (measurement only)."""
        _lib.require_cuda(points, point_offsets)
        if points.dtype != torch.float32 or point_offsets.dtype != torch.int32:
            raise _lib.LidarHipError("points must be float32 and point_offsets int32")
        if points.dim() != 2 or points.shape[1] != self.C:
            raise _lib.LidarHipError(f"points must be (N, {self.C})")
        batch = point_offsets.numel() - 1
        n_max = max(int(n_max), 1)
        self.poll_error()
        ws, nbytes, _ = self._workspace(batch, n_max, points.device)
        if out is None:
            out = self.alloc_outputs(batch, points.device)
        L = _lib.lib()
        algo = 4 if (resident and compact and self.algo in (0, 1, 3, 4) and n_max <= 32768) else self.algo
        hoff = None
        if host_offsets is not None:
            if len(host_offsets) != batch + 1:
                raise _lib.LidarHipError("host_offsets must hold batch + 1 entries")
            hv = [int(v) for v in host_offsets]
            # the LDS-binned launches read the points through THESE offsets (kernel arguments), not the device array: a stale
            # or foreign list would walk past `points`.  Everything needed to refuse it is on the host — no synchronisation.
            if hv[0] < 0 or hv[-1] > points.shape[0] or any(b < a for a, b in zip(hv, hv[1:])):
                raise _lib.LidarHipError(f"host_offsets must be non-decreasing within [0, {points.shape[0]}] (got {hv[0]}..{hv[-1]})")
            if batch and max(b - a for a, b in zip(hv, hv[1:])) > n_max:
                raise _lib.LidarHipError(f"host_offsets hold a frame longer than n_max = {n_max}")
            hoff = (ctypes.c_int * (batch + 1))(*hv)
        if timer is not None:
            L.lidar_voxelize_time_next(timer)
        _lib.check(L.lidar_voxelize_hostoff(_lib.ptr(points), _lib.ptr(point_offsets), hoff, batch, n_max, self.C, self._range_h,
                                            self._vs_h, self._grid_h, self.max_num_points, self.max_voxels, int(bool(compact)),
                                            algo, _lib.ptr(out["voxels"]), _lib.ptr(out["voxel_coords"]),
                                            _lib.ptr(out["voxel_num_points"]), _lib.ptr(out["voxel_offsets"]), _lib.ptr(ws),
                                            nbytes, _lib.stream()), "lidar_voxelize")
        return out

    def error_flag(self, batch, n_max, device):
        """Host-synchronous read of the LDS-binned path's sticky overflow flag (0 = fine)."""
        ws, nbytes, _ = self._workspace(batch, max(int(n_max), 1), device)
        return _lib.lib().lidar_voxelize_error_flag(_lib.ptr(ws), nbytes, batch, max(int(n_max), 1), self.max_voxels)

    def check_error_flag(self, batch, n_max, device):
        flag = self.error_flag(batch, n_max, device)
        if flag != 0:
            raise _lib.LidarHipError(f"lidar_voxelize: LDS hash-bin overflow (flag {flag}); use algo=2")

    def voxelize_frames(self, frames, device="cuda"):
        """Convenience: list of (N_f, C) numpy/torch frames -> compact collated tensors, sliced to
        the true row count (one host sync), as the reference's collate_batch would hand them over."""
        ts = [torch.as_tensor(f, dtype=torch.float32) for f in frames]
        sizes = [int(t.shape[0]) for t in ts]
        offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32)
        pts = torch.cat(ts, 0).to(device).contiguous() if sum(sizes) > 0 else torch.zeros((1, self.C), device=device)
        n_max = max(sizes) if sizes else 1
        offs_d = offs.to(device)
        out = self(pts, offs_d, n_max, compact=True)
        total = int(out["voxel_offsets"][-1].item())
        if self.algo != 2 and self.error_flag(len(sizes), n_max, pts.device) != 0:
            # a hash bin overflowed its LDS budget (thousands of distinct voxels in one bin: adversarial input):
            # redo the batch on the global-hash path, which has no such limit (the sticky flag is cleared by re-init)
            self._ws = {}
            keep, self.algo = self.algo, 2
            try:
                out = self(pts, offs_d, n_max, compact=True)
                total = int(out["voxel_offsets"][-1].item())
            finally:
                self.algo = keep
                self._ws = {}
        return {"voxels": out["voxels"][:total], "voxel_coords": out["voxel_coords"][:total],
                "voxel_num_points": out["voxel_num_points"][:total], "voxel_offsets": out["voxel_offsets"]}
