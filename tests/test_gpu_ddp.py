"""DDP training smoke on the GPU (RCCL, world size 1): one optimisation step of a sparse backbone + RoI-aware pooling head under
torch.nn.parallel.DistributedDataParallel — the reference's training wrapper (tools/train.py:141-142,
pcdet/utils/common_utils.py:170-184).  What it checks: the custom autograd Functions on the hot path (SparseConvFunction:
implicit-GEMM dgrad + MFMA wgrad; RoIAwarePool3dFunction; the stacked grouping backward) produce gradients through DDP's
reducer hooks and bucketed all-reduce that equal the gradients of the same step without DDP, for every parameter.
(The N > 1 path is replicas + the gradient all-reduce only — SURVEY.md §8e — and is covered on CPU by tests/test_dist_gloo.py.)"""
import socket

import numpy as np
import pytest
import torch
import torch.nn as nn

from lidardetection_amd import spconv, synth
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone
from lidardetection_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as smod
from lidardetection_amd.pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils
from lidardetection_amd.pcdet.utils.cfg import AttrDict

pytestmark = pytest.mark.gpu


class _TinyDetector(nn.Module):
    """sparse backbone -> voxel centres as points -> (a) RoI-aware max pooling of the conv3 features, (b) a stacked SA module"""

    def __init__(self, grid):
        super().__init__()
        self.grid = grid
        self.backbone = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, grid)
        self.pool = roiaware_pool3d_utils.RoIAwarePool3d(out_size=4, max_pts_each_voxel=32)
        self.sa = smod.StackSAModuleMSG(radii=[2.0], nsamples=[8], mlps=[[64, 16]], use_xyz=True, pool_method='max_pool')
        self.fc = nn.Linear(64, 1)

    def forward(self, feats, coords, rois, batch_size):
        bd = self.backbone({"voxel_features": feats, "voxel_coords": coords, "batch_size": batch_size})
        t = bd["multi_scale_3d_features"]["x_conv3"]
        centres = (t.indices[:, [3, 2, 1]].float() + 0.5) * 4.0                       # stride-4 voxel centres, grid units
        sel = t.indices[:, 0] == 0
        pooled = self.pool(rois, centres[sel].contiguous(), t.features[sel].contiguous(), pool_method='max')   # (R, 4, 4, 4, 64)
        cnt = torch.bincount(t.indices[:, 0].long(), minlength=batch_size).int()
        new_xyz = centres[::7].contiguous()
        new_cnt = torch.bincount(t.indices[::7, 0].long(), minlength=batch_size).int()
        _, sa = self.sa(centres.contiguous(), cnt, new_xyz, new_cnt, t.features.contiguous())
        return self.fc(pooled).pow(2).mean() + sa.pow(2).mean() + bd["encoded_spconv_tensor"].features.pow(2).mean()


def _inputs(dev):
    grid = [48, 40, 24]
    r = np.random.default_rng(9)
    cells, B = 25 * 40 * 48, 2
    pick = np.concatenate([r.choice(cells, 3000, replace=False) + b * cells for b in range(B)])
    b_, rem = np.divmod(pick, cells)
    z, rem = np.divmod(rem, 40 * 48)
    y, x = np.divmod(rem, 48)
    coords = torch.from_numpy(np.stack([b_, z, y, x], 1).astype(np.int32)).to(dev)
    feats = torch.from_numpy(r.standard_normal((len(pick), 4)).astype(np.float32)).to(dev)
    rois = torch.tensor([[20.0, 20.0, 12.0, 16.0, 12.0, 10.0, 0.3], [30.0, 10.0, 8.0, 10.0, 14.0, 8.0, -0.7]], device=dev)
    return grid, feats, coords, rois, B


def test_ddp_step_equals_plain_step_through_the_custom_autograd_functions(dev):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    grid, feats, coords, rois, B = _inputs(dev)
    torch.manual_seed(4)
    plain = _TinyDetector(grid).to(dev).train()
    torch.manual_seed(4)
    wrapped = _TinyDetector(grid).to(dev).train()
    plain(feats, coords, rois, B).backward()
    ref = {n: p.grad.clone() for n, p in plain.named_parameters() if p.grad is not None}
    assert len(ref) == sum(1 for _ in plain.parameters())                 # every parameter takes part in the loss
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)     # RCCL
    try:
        ddp = DistributedDataParallel(wrapped, device_ids=[dev.index or 0], bucket_cap_mb=1)         # several buckets
        opt = torch.optim.SGD(ddp.parameters(), lr=1e-3)
        loss = ddp(feats, coords, rois, B)
        loss.backward()
        torch.cuda.synchronize()
        for n, p in wrapped.named_parameters():
            assert p.grad is not None, n
            scale = max(1.0, float(ref[n].abs().max()))
            # same kernels, deterministic summation orders (sparse conv, wgrad); float atomics only in the pooling / grouping backward
            assert float((p.grad - ref[n]).abs().max()) <= 1e-5 * scale, n
        before = {n: p.detach().clone() for n, p in wrapped.named_parameters()}
        opt.step()
        assert any(not torch.equal(before[n], p) for n, p in wrapped.named_parameters())
        assert all(torch.isfinite(p).all() for p in wrapped.parameters())
    finally:
        dist.destroy_process_group()
