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


def _run_bench(extra_args, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *extra_args], env=e, capture_output=True, text=True,
                          timeout=300)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no torch.distributed launcher) must start two ranks and print ONE line with n_gpus 2
    (reference launch: tools/scripts/dist_train.sh:7, pcdet/utils/common_utils.py:170-184).  --dry-run = gloo, no kernels."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["dry_run"] is True and rec["frames_per_rank"] == 16


def test_bench_refuses_a_world_size_mismatch():
    r = _run_bench(["--gpus", "4", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_train_ddp_mode_two_ranks_dry_run():
    """`bench.py --gpus 2 --mode train-ddp` (BASELINE configs[4]'s DDP leg; reference: tools/train.py:141-142 wraps the model in
    DistributedDataParallel, process group from pcdet/utils/common_utils.py:170-184): two ranks through the launcher, the DDP
    reducer, the no_sync re-timing, the stand-alone all-reduce and the single JSON line — on CPU / gloo with the stand-in model."""
    import json
    r = _run_bench(["--gpus", "2", "--mode", "train-ddp", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["dry_run"] is True and rec["scaling"] == "weak"
    assert rec["config"]["batch_per_gpu"] == 4 and rec["config"]["parallelism"].startswith("ddp2")
    ar = rec["allreduce"]
    assert ar["gradient_bytes"] > 0 and ar["alone_ms"] > 0 and ar["step_ms_no_sync"] > 0
    assert abs(rec["value"] - 4 * 2 * 3 / (rec["ms_per_step"] * 3e-3)) < 1e-6 * rec["value"]       # whole-job samples / s


def test_gemm_choice_export_import_round_trip():
    """N > 1: rank 0's library-GEMM picks travel to the other ranks as a flat int list (csrc/dense_gemm.hip export / import);
    on a box without a GPU there are no plans yet: empty export, import of a foreign list accepted (it only seeds later plans)"""
    assert dist_utils.export_gemm_choices() == []
    dist_utils.import_gemm_choices([857088 & 0x7FFFFFFF, 0, 64, 128, 384, 3, 2])
    dist_utils.import_gemm_choices([])
