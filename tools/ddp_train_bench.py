"""BASELINE.json configs[4], the DDP leg: SECOND-MultiHead NuScenes, bs 4 per GPU, one TRAINING step = forward + loss + backward
(gradients bucket-all-reduced by torch DistributedDataParallel over RCCL / xGMI) + optimiser step — what the reference runs as
`tools/train.py --launcher pytorch` (tools/train.py:141-142 wraps the model in DistributedDataParallel; process group from
pcdet/utils/common_utils.py:170-184; sharded sampler pcdet/datasets/__init__.py:26-46).  Called by `bench.py --mode train-ddp`.

What runs: HIP voxelise (no grad) -> MeanVFE -> VoxelResBackBone8x in train mode (module sequence: SparseConvFunction forward /
implicit-GEMM dgrad / MFMA wgrad, train-mode BatchNorm1d) -> dense() -> BaseBEVBackbone + shared conv + the six multi-head branches
(stock torch modules, MIOpen) -> AnchorHeadMulti-shaped loss: sigmoid focal classification loss (alpha 0.25, gamma 2) + weighted
smooth-L1 box loss (beta 1/9) over the positive anchors (pcdet/utils/loss_utils.py:9-134), against synthetic targets (there is no
dataset here: 1 % of anchors positive, random residuals) -> backward -> AdamW step.  Parallelism: replicas + the gradient
all-reduce, nothing else.

Prints ONE JSON line on rank 0.  `value` = training samples / s of the whole job (all ranks), clocked barrier-to-barrier, max over
ranks.  `allreduce`: the step re-timed under `no_sync()` (same kernels, no collective) -> exposed_ms = what the all-reduce adds to a
step after DDP's overlap with backward; alone_ms = one all-reduce of the flat gradient size by itself; bytes = gradient bytes per
step.  With one rank the collective is a no-op copy and says so.  `--dry-run` runs the same control flow on CPU / gloo with a
small stand-in model (tests/test_dist_gloo.py): launcher, process group, DDP hooks, no_sync, timing, the JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from lidardetection_amd import dist_utils  # noqa: E402


def sigmoid_focal_loss(logits, targets, alpha=0.25, gamma=2.0):
    """SigmoidFocalClassificationLoss (pcdet/utils/loss_utils.py:9-67): per-element, unnormalised"""
    p = torch.sigmoid(logits)
    alpha_w = targets * alpha + (1 - targets) * (1 - alpha)
    pt = targets * (1.0 - p) + (1.0 - targets) * p
    bce = torch.clamp(logits, min=0) - logits * targets + torch.log1p(torch.exp(-torch.abs(logits)))
    return alpha_w * torch.pow(pt, gamma) * bce


def smooth_l1(diff, beta=1.0 / 9.0):
    """WeightedSmoothL1Loss (loss_utils.py:70-134)"""
    n = torch.abs(diff)
    return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)


class TrainStepModel(nn.Module):
    """the detector's trainable part with the loss inside forward(), so that DistributedDataParallel wraps exactly one module"""

    def __init__(self, det):
        super().__init__()
        self.det = det

    def forward(self, feats, coords, cls_t, box_t, pos):
        d = self.det
        bd = d.backbone3d({"voxel_features": feats, "voxel_coords": coords, "batch_size": d.B})
        t = bd["encoded_spconv_tensor"]
        x = t.dense()                                                    # differentiable path (index_put), (B, C, D, H, W)
        B, C, D, H, W = x.shape
        x = x.view(B, C * D, H, W)
        ups = []
        for blk, de in zip(d.blocks, d.deblocks):
            x = blk(x)
            ups.append(de(x))
        heads = d.heads_reference_layout(torch.cat(ups, dim=1))
        loss = 0.0
        for (cls, box), ct, bt, pm in zip(heads, cls_t, box_t, pos):
            npos = pm.sum().clamp(min=1.0)
            loss = loss + sigmoid_focal_loss(cls, ct).sum() / npos + 2.0 * (smooth_l1(box - bt).sum(-1) * pm).sum() / npos
        return loss


class DryModel(nn.Module):
    """stand-in for --dry-run (CPU / gloo): same call signature, a few small layers"""

    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(32, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, 8))

    def forward(self, feats, coords, cls_t, box_t, pos):
        return self.net(feats).square().mean()


def run(args, rank, local, world):
    dry = bool(args.dry_run)
    backend = "gloo" if dry else "nccl"
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dry:
        torch.cuda.set_device(local)
    if not dist.is_initialized():                     # also for one rank: DDP needs a process group (the all-reduce is then a copy)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    device = torch.device("cpu") if dry else torch.device("cuda", local)
    B = 4 if args.batch == 16 else args.batch         # bench.py's default --batch is the PointPillar one; this config is bs 4 / GPU
    torch.manual_seed(0)                              # same initial weights on every rank (DDP broadcasts rank 0's anyway)
    if dry:
        model = DryModel()
        g = torch.Generator().manual_seed(100 + rank)
        inputs = (torch.randn(B * 64, 32, generator=g), None, None, None, None)
        cfg_name = "dry-run stand-in (CPU, gloo)"
    else:
        from lidardetection_amd import pillar_ops, synth
        from lidardetection_amd.second_multihead import SECONDMultiHeadNuScenes
        torch.backends.cudnn.benchmark = os.environ.get("LIDAR_BENCH_MIOPEN_FIND", "0") != "0"    # heuristic pick: same on every rank
        frames = [synth.cloud_nus(4000 + rank * B + f) for f in range(B)]
        sizes = [len(f) for f in frames]
        pts = torch.from_numpy(np.concatenate(frames, 0)).to(device)
        offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=device)
        det = SECONDMultiHeadNuScenes(batch_size=B, n_max=max(sizes), device=device)
        det.train()
        with torch.no_grad():                         # voxelisation + MeanVFE: the data side, no parameters
            feats, coords = det.voxelize_vfe(pts, offs)
            feats, coords = feats.clone(), coords.clone()
        model = TrainStepModel(det)
        g = torch.Generator(device="cpu").manual_seed(7 + rank)
        cls_t, box_t, pos = [], [], []
        H, W = det.grid[1] // 8, det.grid[0] // 8
        for head in det.rpn_heads:
            n = head.A * H * W
            pm = (torch.rand(B, n, generator=g) < 0.01).float()
            lab = torch.randint(0, head.num_class, (B, n), generator=g)
            cls_t.append((F.one_hot(lab, head.num_class).float() * pm.unsqueeze(-1)).to(device))
            box_t.append((torch.randn(B, n, head.code_size, generator=g) * 0.2).to(device))
            pos.append(pm.to(device))
        inputs = (feats, coords, cls_t, box_t, pos)
        cfg_name = "SECOND-MultiHead NuScenes (cbgs_second_multihead.yaml), VoxelResBackBone8x, 6 heads / 10 classes"
    from torch.nn.parallel import DistributedDataParallel
    ddp = DistributedDataParallel(model, device_ids=None if dry else [local], bucket_cap_mb=25)
    opt = torch.optim.AdamW(ddp.parameters(), lr=3e-3, weight_decay=0.01)
    params = [p for p in ddp.parameters() if p.requires_grad]
    grad_bytes = sum(p.numel() * p.element_size() for p in params)

    def sync():
        if not dry:
            torch.cuda.synchronize()
        dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    def step(no_sync=False):
        opt.zero_grad(set_to_none=True)
        if no_sync:
            with ddp.no_sync():
                loss = ddp(*inputs)
                loss.backward()
        else:
            loss = ddp(*inputs)
            loss.backward()
        opt.step()
        return loss

    def timed(n, no_sync=False):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step(no_sync)
        sync()
        dt = time.perf_counter() - t0
        return dist_utils.max_over_ranks(dt, dist, device), float(loss.detach())

    for _ in range(max(args.warmup, 1)):
        step()
    # A / B / A / B, the faster of each pair: on one GPU the two are the same work and first-run effects would otherwise show up
    # as a (negative) "all-reduce" time
    dt, loss = timed(args.steps)
    dt_ns, _ = timed(args.steps, no_sync=True)
    dt2, loss = timed(args.steps)
    dt_ns2, _ = timed(args.steps, no_sync=True)
    dt, dt_ns = min(dt, dt2), min(dt_ns, dt_ns2)
    # one all-reduce of the gradient volume by itself (what the buckets move per step)
    flat = torch.zeros(grad_bytes // 4, dtype=torch.float32, device=device)
    for _ in range(2):
        dist.all_reduce(flat)
    sync()
    t0 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(flat)
    sync()
    ar_alone = dist_utils.max_over_ranks((time.perf_counter() - t0) / 5, dist, device)
    if rank == 0:
        step_ms, ns_ms = dt / args.steps * 1e3, dt_ns / args.steps * 1e3
        print(json.dumps({
            "metric": "training samples/sec SECOND-MultiHead NuScenes DDP (fwd + loss + bwd + all-reduce + AdamW)", "value": B * world * args.steps / dt,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": step_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": dry,
            "config": {"workload": cfg_name, "batch_per_gpu": B, "parallelism": f"ddp{world} (replicas + bucketed gradient all-reduce, RCCL)",
                       "bucket_cap_mb": 25, "optimizer": "AdamW"},
            "allreduce": {"gradient_bytes": grad_bytes, "step_ms_no_sync": ns_ms, "exposed_ms": step_ms - ns_ms,
                          "exposed_share_of_step": (step_ms - ns_ms) / step_ms, "alone_ms": ar_alone * 1e3,
                          "alone_bus_GBs": (2 * (world - 1) / world * grad_bytes / ar_alone / 1e9) if world > 1 else None,
                          "note": "one rank: the collective is a local copy" if world == 1 else "ring all-reduce over xGMI (RCCL)"},
            "loss": loss, "hardware_status": "unmeasured on N > 1 GPUs until a SCALE record exists" if world == 1 else "measured"}))
    dist.barrier()
    dist.destroy_process_group()
