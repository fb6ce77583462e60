"""Pinned double-buffered feed of raw point clouds to the device.

The reference moves a collated batch to the GPU synchronously, key by key (pcdet/models/__init__.py:16-22:
`torch.from_numpy(val).float().cuda()` from pageable memory) after its DataLoader workers have voxelised every frame on the CPU
and `collate_batch` has concatenated voxels / coordinates / counts (pcdet/datasets/dataset.py:153-185).  With the voxeliser on the
device the only thing that has to cross PCIe is the RAW cloud — 16 B per point, 5 MB for 16 KITTI frames instead of the 137 MB
padded voxel buffer — and it can cross while the previous batch is still being processed:

    feeder.submit(frames_of_batch_k+1)     # host: copy into pinned staging, enqueue H2D on the copy stream, return at once
    batch = feeder.get()                   # compute stream waits (on the device, not the host) for batch k's copy
    model(batch.points, batch.offsets, batch.host_offsets)

`depth` staging slots (host pinned + device) rotate; a slot's device buffer is overwritten only after everything that was enqueued
on the caller's stream before the overwriting `submit()` has finished (the slot's readers were enqueued `depth` batches earlier), a
slot's pinned buffer only after its own copy has completed.  No host synchronisation in steady state.
"""
import collections

import numpy as np
import torch

from . import _lib

FedBatch = collections.namedtuple("FedBatch", "points offsets host_offsets n_max batch")


class PinnedPointFeeder:
    def __init__(self, max_points, num_features=4, max_batch=16, device="cuda", depth=2):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LidarHipError("PinnedPointFeeder feeds a GPU; there is no CPU path")
        self.C, self.max_points, self.max_batch, self.depth = int(num_features), int(max_points), int(max_batch), int(depth)
        if self.depth < 2:
            raise _lib.LidarHipError("depth >= 2: one slot is copied into while the other is read")
        self._hpts = [torch.empty((self.max_points, self.C), dtype=torch.float32).pin_memory() for _ in range(self.depth)]
        self._hoff = [torch.empty((self.max_batch + 1,), dtype=torch.int32).pin_memory() for _ in range(self.depth)]
        self._hpts_np = [t.numpy() for t in self._hpts]
        self._hoff_np = [t.numpy() for t in self._hoff]
        self._dpts = [torch.empty((self.max_points, self.C), dtype=torch.float32, device=self.device) for _ in range(self.depth)]
        self._doff = [torch.empty((self.max_batch + 1,), dtype=torch.int32, device=self.device) for _ in range(self.depth)]
        self._copy_stream = torch.cuda.Stream(self.device)
        self._copied = [torch.cuda.Event() for _ in range(self.depth)]      # slot's H2D done (copy stream)
        self._free = [torch.cuda.Event() for _ in range(self.depth)]        # "the slot's readers are done" marks (caller's stream)
        self._handed = [False] * self.depth
        self._meta = [None] * self.depth
        self._head = self._tail = 0                                         # submitted / handed out

    def pending(self):
        return self._head - self._tail

    def submit(self, frames):
        """frames: list of (N_f, C) float32 arrays (numpy, or CPU tensors) — one batch, as collate_batch receives it.  Copies them
        back to back into the slot's pinned buffer and enqueues the H2D copy.  Raises if every slot is still un-fetched."""
        if self.pending() >= self.depth:
            raise _lib.LidarHipError("PinnedPointFeeder: every staging slot holds an un-fetched batch (call get() first)")
        sizes = [int(f.shape[0]) for f in frames]
        total, B = sum(sizes), len(frames)
        if B > self.max_batch or total > self.max_points:
            raise _lib.LidarHipError(f"PinnedPointFeeder: batch of {B} frames / {total} points exceeds ({self.max_batch}, {self.max_points})")
        slot = self._head % self.depth
        self._copied[slot].synchronize()             # the pinned buffer's previous copy has left it (long ago in steady state)
        dst, a = self._hpts_np[slot], 0
        for f, n in zip(frames, sizes):
            f = f.numpy() if isinstance(f, torch.Tensor) else np.asarray(f)
            if f.ndim != 2 or f.shape[1] != self.C:
                raise _lib.LidarHipError(f"PinnedPointFeeder: frames must be (N, {self.C})")
            dst[a:a + n] = f                          # converts to float32 like the reference's .float()
            a += n
        hoffs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
        self._hoff_np[slot][:B + 1] = hoffs
        cs = self._copy_stream
        if self._handed[slot]:
            # the device buffer was handed out `depth` batches ago: whatever reads it was enqueued on the caller's stream before
            # this call, so a mark recorded there NOW bounds its last reader (the copy waits on the device, the host does not)
            self._free[slot].record(torch.cuda.current_stream(self.device))
            cs.wait_event(self._free[slot])
            self._handed[slot] = False
        with torch.cuda.stream(cs):
            self._dpts[slot][:max(total, 1)].copy_(self._hpts[slot][:max(total, 1)], non_blocking=True)
            self._doff[slot][:B + 1].copy_(self._hoff[slot][:B + 1], non_blocking=True)
            self._copied[slot].record(cs)
        self._meta[slot] = ([int(v) for v in hoffs], max(sizes) if sizes else 0, B, total)
        self._head += 1

    def get(self):
        """-> FedBatch(points (sum N, C) device view, offsets (B+1) int32 device, host_offsets list, n_max, batch) of the oldest
        submitted batch; the CURRENT stream waits for its copy on the device.  The views stay valid until `depth` further batches
        have been submitted."""
        if self.pending() == 0:
            raise _lib.LidarHipError("PinnedPointFeeder.get(): nothing submitted")
        slot = self._tail % self.depth
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._copied[slot])
        self._handed[slot] = True
        hoffs, n_max, B, total = self._meta[slot]
        self._tail += 1
        return FedBatch(self._dpts[slot][:max(total, 1)] if total else self._dpts[slot][:0], self._doff[slot][:B + 1], hoffs, n_max, B)
