"""Multi-GPU plumbing for the replica-parallel hot path: one process per GPU, frames sharded
round-robin exactly like the reference's eval sampler (pcdet/datasets/__init__.py:26-46: indices[rank::world], padded by
wrapping so every rank gets the same count), no data-path collective; torch.distributed (RCCL on ROCm, gloo on CPU)
is used only for barriers and to agree on the slowest rank's clock."""
import math
import os

import torch


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (no-op for a single process)."""
    rank, local, world = env_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def shard_indices(num_items, rank, world):
    """Round-robin shard of range(num_items) with wrap-around padding (every rank gets ceil(n / world) items)."""
    if world <= 1:
        return list(range(num_items))
    per = int(math.ceil(num_items / world))
    idx = list(range(num_items))
    idx += idx[: per * world - len(idx)]
    return idx[rank: per * world: world]


def max_over_ranks(value, dist=None, device="cpu"):
    """The slowest rank's value (what the throughput of a replica-parallel job is measured against)."""
    if dist is None:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(dist=None, cuda=False):
    if cuda:
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    if cuda:
        torch.cuda.synchronize()


def export_gemm_choices():
    """rank 0's library-GEMM candidate picks (csrc/dense_gemm.hip) as a flat list of ints (7 per shape) for a broadcast"""
    import ctypes
    from . import _lib
    L = _lib.lib()
    n = L.lidar_dense_gemm_export_choices(None, 0)
    buf = (ctypes.c_int * (7 * max(n, 1)))()
    n = min(n, L.lidar_dense_gemm_export_choices(buf, max(n, 1)))
    return [int(v) for v in buf[:7 * n]]


def import_gemm_choices(flat):
    """the other ranks, before their first call of those shapes: run the kernels rank 0 kept instead of timing their own"""
    import ctypes
    from . import _lib
    if not flat:
        return
    buf = (ctypes.c_int * len(flat))(*flat)
    _lib.check(_lib.lib().lidar_dense_gemm_import_choices(buf, len(flat) // 7), "lidar_dense_gemm_import_choices")
