"""World-size-2 gloo test (CPU) of the replica-parallel plumbing used by bench.py --gpus N: frame sharding identical to the
reference's eval DistributedSampler, barrier, and max-over-ranks clock."""
import os
import socket

import torch
import torch.multiprocessing as mp

from lidardetection_amd import dist_utils


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = dist_utils.init_from_env("gloo")
    shard = dist_utils.shard_indices(n_frames, rank, world)
    dist_utils.barrier(dist)
    slowest = dist_utils.max_over_ranks(1.0 + rank, dist)       # rank r "took" 1+r seconds
    total = torch.tensor([float(len(shard))], dtype=torch.float64)
    dist.all_reduce(total)
    q.put((rank, shard, slowest, float(total.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_clock():
    world, n = 2, 37
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, slow0, tot0), (r1, s1, slow1, tot1) = res
    assert s0 == list(range(0, 38, 2))[:19] and len(s0) == len(s1) == 19        # ceil(37/2), padded by wrap-around
    assert s1[:-1] == list(range(1, 37, 2)) and s1[-1] == 0
    assert sorted(set(s0 + s1)) == list(range(n))                                # every frame covered
    assert slow0 == slow1 == 2.0 and tot0 == tot1 == 38.0


def test_single_process_is_identity():
    assert dist_utils.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    assert dist_utils.max_over_ranks(3.5) == 3.5
