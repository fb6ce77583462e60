"""The callers of a22 / a23 / a26 and the inverse convolutions at the shapes the reference's configs use (VERDICT r02 item 5;
until r03 these ops were parity-tested on toy scenes only and timed at full size in tools/pvrcnn_ops_bench.py):
  * RoI-aware pooling, Part-A2 head: 128 RoIs, 12^3 grid, <= 128 points per voxel, ~16 k points, C = 128 (max) and C = 4 (avg)
    (tools/cfgs/kitti_models/PartA2.yaml:126-129, pcdet/models/roi_heads/partA2_head.py:53-56,138-143)
  * RoI-point pooling, PointRCNN head: 16 384 points, 128 boxes, 512 samples, C = 130 (pcdet/models/roi_heads/pointrcnn_head.py:101-119)
  * pointnet2_batch set-abstraction chain of the PointRCNN backbone: 16 384 -> 4 096 -> 1 024 -> 256 -> 64 points, two radii per
    level (pcdet/models/backbones_3d/pointnet2_backbone.py:27-46, tools/cfgs/kitti_models/pointrcnn.yaml)
  * UNetV2's decoder: three SparseInverseConv3d (spconv4 -> spconv3 -> spconv2) on one full-size 41 x 1600 x 1408 SECOND frame
    (pcdet/models/backbones_3d/spconv_unet.py:113-123)
all against the existing CPU oracles (oracle/src/points_oracle.c, oracle/spconv_sparse_oracle.py); integer outputs exact, fp32
features to 1e-4 of their scale.  The measured GPU time of every op is printed (and lands in profiles/r03/ through the log)."""
import time

import numpy as np
import pytest
import torch

from lidardetection_amd import pillar_ops, spconv, synth
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone
from lidardetection_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as butils
from lidardetection_amd.pcdet.ops.roiaware_pool3d import roiaware_pool3d_utils
from lidardetection_amd.pcdet.ops.roipoint_pool3d import roipoint_pool3d_utils
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.voxelizer import BatchVoxelizer
from oracle import c_oracle, spconv_sparse_oracle as sp

pytestmark = pytest.mark.gpu


def _gpu_ms(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def _kitti_scene(seed, nbox, npts):
    """`npts` points of a ring cloud and `nbox` car / pedestrian / cyclist-sized boxes centred on points of it; points whose
    in-box decision could flip with the last ulp of cos / sin (membership differs between boxes shrunk / grown by 1e-4, in
    float64) are removed — their number is bounded at 0.5 %"""
    r = np.random.default_rng(seed)
    cloud = np.concatenate([synth.cloud_ring(2000 + seed), synth.cloud_ring(2100 + seed)], 0)[:, :3]
    cloud = cloud[r.permutation(len(cloud))]
    ctr = cloud[r.choice(len(cloud), nbox, replace=False)] + r.normal(0, 0.3, (nbox, 3))
    size = np.array([[3.9, 1.6, 1.56], [0.8, 0.6, 1.73], [1.76, 0.6, 1.73]])[r.integers(0, 3, nbox)] * r.uniform(0.9, 1.3, (nbox, 3))
    boxes = np.concatenate([ctr, size, r.uniform(-np.pi, np.pi, (nbox, 1))], 1).astype(np.float32)
    b, q = boxes.astype(np.float64), cloud.astype(np.float64)
    d = q[None, :, :] - b[:, None, :3]
    c, s_ = np.cos(-b[:, 6])[:, None], np.sin(-b[:, 6])[:, None]
    lx, ly = np.abs(d[..., 0] * c - d[..., 1] * s_), np.abs(d[..., 0] * s_ + d[..., 1] * c)
    inside = lambda m: (lx < b[:, None, 3] / 2 + m) & (ly < b[:, None, 4] / 2 + m) & (np.abs(d[..., 2]) < b[:, None, 5] / 2 + m)
    amb = (inside(1e-4) != inside(-1e-4)).any(0)
    assert amb.sum() <= 0.005 * len(cloud), amb.sum()
    pts = cloud[~amb][:npts].astype(np.float32)
    assert len(pts) == npts
    return boxes, np.ascontiguousarray(pts)


@pytest.mark.parametrize("C,method", [(128, "max"), (4, "avg")])
def test_roiaware_pool3d_parta2_shapes(dev, C, method):
    boxes, pts = _kitti_scene(1, 128, 16384)
    feat = np.random.default_rng(3).standard_normal((len(pts), C)).astype(np.float32)
    pooled_o, argmax_o, pidx_o = c_oracle.roiaware_pool3d(boxes, pts, feat, (12, 12, 12), 128, 0 if method == "max" else 1)
    tb, tp, tf = torch.from_numpy(boxes).to(dev), torch.from_numpy(pts).to(dev), torch.from_numpy(feat).to(dev).requires_grad_(True)
    fn = roiaware_pool3d_utils.RoIAwarePool3dFunction
    pooled = fn.apply(tb, tp, tf, 12, 128, method)
    pidx, argmax, _, _, _ = pooled.grad_fn.roiaware_pool3d_for_backward
    assert pooled.shape == (128, 12, 12, 12, C)
    assert np.array_equal(pidx.cpu().numpy(), pidx_o)                                  # per-voxel point lists, exact
    assert (pidx_o[..., 0] > 0).sum() > 2000
    if method == "max":
        assert np.array_equal(argmax.cpu().numpy(), argmax_o) and np.array_equal(pooled.detach().cpu().numpy(), pooled_o)
    else:
        np.testing.assert_allclose(pooled.detach().cpu().numpy(), pooled_o, rtol=0, atol=1e-5)
    go = np.random.default_rng(4).standard_normal(pooled.shape).astype(np.float32)
    pooled.backward(torch.from_numpy(go).to(dev))
    gi_o = c_oracle.roiaware_pool3d_backward(pidx_o, argmax_o, go, len(pts), 0 if method == "max" else 1)
    np.testing.assert_allclose(tf.grad.cpu().numpy(), gi_o, rtol=1e-4, atol=1e-4)
    with torch.no_grad():
        ms = _gpu_ms(lambda: fn.apply(tb, tp, tf.detach(), 12, 128, method))
    print(f"[ref shapes] roiaware_pool3d 128 x 12^3 x 128, {len(pts)} pts, C {C} ({method}): {ms * 1e3:.0f} us")


def test_roipoint_pool3d_pointrcnn_shapes(dev):
    B, N, M, C, S = 2, 16384, 128, 130, 512
    scenes = [_kitti_scene(5 + b, M, N) for b in range(B)]
    bx = np.stack([s[0] for s in scenes], 0)
    xyz = np.stack([s[1] for s in scenes], 0)
    feat = np.random.default_rng(6).standard_normal((B, N, C)).astype(np.float32)
    pool = roipoint_pool3d_utils.RoIPointPool3d(num_sampled_points=S, pool_extra_width=[0.0, 0.0, 0.0])   # (the 1e-4 margin is on the plain boxes)
    t = lambda a: torch.from_numpy(a).to(dev)
    pooled, empty = pool(t(xyz), t(feat), t(bx))
    po, eo = c_oracle.roipoint_pool3d(xyz, bx, feat, S)
    assert pooled.shape == (B, M, S, 3 + C)
    assert np.array_equal(empty.cpu().numpy(), eo)
    assert np.array_equal(pooled.cpu().numpy(), po)
    assert (eo == 0).sum() > B * M // 2
    ms = _gpu_ms(lambda: pool(t(xyz), t(feat), t(bx)))
    print(f"[ref shapes] roipoint_pool3d {B} x {N} pts, {M} boxes, {S} samples, C {C}: {ms * 1e3:.0f} us (incl. the H2D copies of the call)")


def test_pointnet2_batch_sa_chain_pointrcnn_shapes(dev):
    """FPS -> gather -> ball query (two radii) -> grouping at every level of the PointRCNN backbone, level by level on the
    oracle's own centres (so an index mismatch cannot hide behind the next level)"""
    B, N = 2, 16384
    xyz = np.stack([np.concatenate([synth.cloud_ring(2000 + k), synth.cloud_ring(2050 + k)], 0)[:N, :3] for k in range(B)], 0)
    xyz = np.ascontiguousarray(xyz, np.float32)
    r = np.random.default_rng(12)
    feat = r.standard_normal((B, 8, N)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    times = []
    for m, radii, ns in ((4096, (0.1, 0.5), (16, 32)), (1024, (0.5, 1.0), (16, 32)), (256, (1.0, 2.0), (16, 32)), (64, (2.0, 4.0), (16, 32))):
        n = xyz.shape[1]
        txyz = t(xyz)
        fi = butils.furthest_point_sample(txyz, m)
        fi_o = c_oracle.fps(xyz, m)
        assert np.array_equal(fi.cpu().numpy(), fi_o), f"FPS {n} -> {m}"
        times.append((f"fps {n}->{m}", _gpu_ms(lambda: butils.furthest_point_sample(txyz, m), n=2)))
        new = np.ascontiguousarray(np.stack([xyz[b][fi_o[b]] for b in range(B)], 0))
        g_new = butils.gather_operation(txyz.transpose(1, 2).contiguous(), fi).transpose(1, 2).contiguous()
        assert np.array_equal(g_new.cpu().numpy(), new)
        for rad, k in zip(radii, ns):
            idx = butils.ball_query(rad, k, txyz, g_new)
            idx_o = c_oracle.ball_query_batch(rad, k, xyz, new)
            assert np.array_equal(idx.cpu().numpy(), idx_o), f"ball query r {rad} at {n} -> {m}"
            g = butils.grouping_operation(t(feat), idx)
            assert np.array_equal(g.cpu().numpy(), c_oracle.group_points_batch(feat, idx_o))
            times.append((f"ball r{rad} {m}x{n}", _gpu_ms(lambda: butils.ball_query(rad, k, txyz, g_new))))
        feat = np.ascontiguousarray(np.stack([feat[b][:, fi_o[b]] for b in range(B)], 0))
        xyz = new
    print("[ref shapes] pointnet2_batch SA chain 16384->4096->1024->256->64, B 2: " + ", ".join(f"{k} {v * 1e3:.0f} us" for k, v in times))


def test_unet_decoder_inverse_convs_full_second_grid(dev):
    """VoxelBackBone8x's encoder on ONE full-size frame, then UNetV2's three inverse convolutions back up through the encoder's own
    rulebooks (spconv4 -> spconv3 -> spconv2), each + BatchNorm1d + ReLU as post_act_block builds it; every level against the
    sparse fp64 oracle (pinned to the dense transposed-convolution oracle on small grids, tests/test_oracle_pins.py)."""
    from test_gpu_configs import _bn64
    o = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000).voxelize_frames([synth.cloud_ring(2009)], device=dev)
    feats = pillar_ops.mean_vfe(o["voxels"], o["voxel_num_points"])
    torch.manual_seed(7)
    m = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
    norm = lambda c: torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01)
    inv = [spconv_backbone.post_act_block(64, 64, 3, norm_fn=norm, indice_key="spconv4", conv_type="inverseconv"),
           spconv_backbone.post_act_block(64, 32, 3, norm_fn=norm, indice_key="spconv3", conv_type="inverseconv"),
           spconv_backbone.post_act_block(32, 16, 3, norm_fn=norm, indice_key="spconv2", conv_type="inverseconv")]
    for blk in inv:
        blk.to(dev).eval()
    with torch.no_grad():
        for mod in list(m.modules()) + [x for blk in inv for x in blk.modules()]:
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.8, 1.2); mod.bias.uniform_(-0.1, 0.1)
        x = spconv.SparseConvTensor(feats, o["voxel_coords"].int(), m.sparse_shape, 1)
        levels = []
        for name in ("conv_input", "conv1", "conv2", "conv3", "conv4"):
            x = getattr(m, name)(x)
            levels.append(x)
        ups, y = [], levels[4]
        for blk in inv:
            y = blk(y)
            ups.append(y)
        ms = _gpu_ms(lambda: inv[2](inv[1](inv[0](levels[4]))))
    # oracle: the inverse convs take the GPU's own encoder output (x_conv4) as input, so only the decoder is under test here
    f = levels[4].features.cpu().double().numpy()
    idx_small, shape_small = levels[4].indices.cpu().numpy().astype(np.int64), levels[4].spatial_shape
    for blk, target, got, (ks, st, pd) in zip(inv, (levels[3], levels[2], levels[1]), ups,
                                              (((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)))):
        conv, bn = blk[0], blk[1]
        assert (list(conv.kernel_size) if hasattr(conv, "kernel_size") else None) is not None
        idx_orig = target.indices.cpu().numpy().astype(np.int64)
        assert got.spatial_shape == target.spatial_shape and torch.equal(got.indices, target.indices)     # back on the encoder's sites
        want = sp.inverse_conv(f, idx_small, shape_small, idx_orig, target.spatial_shape, conv.weight.detach().cpu().double().numpy(), None,
                               list(ks), list(st), list(pd))
        want = np.maximum(_bn64(bn, want), 0)
        scale = max(1.0, float(np.abs(want).max()))
        err = float(np.abs(got.features.cpu().double().numpy() - want).max()) / scale
        assert err <= 1e-4, err
        f, idx_small, shape_small = want, idx_orig, target.spatial_shape
    print(f"[ref shapes] UNetV2 decoder: 3 SparseInverseConv3d + BN + ReLU on one full SECOND frame ({levels[4].features.shape[0]} -> "
          f"{ups[-1].features.shape[0]} sites): {ms * 1e3:.0f} us")
