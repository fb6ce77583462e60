"""GPU parity of the mirrored model modules that sit on the hot path (PillarVFE, MeanVFE, PointPillarScatter,
HeightCompression, VoxelBackBone8x / VoxelResBackBone8x, class_agnostic_nms / multi_classes_nms) — module-level API as in
the reference (batch_dict in, batch_dict out), checked against the reference-generated golden fixtures and the oracles."""
import os

import numpy as np
import pytest
import torch

from lidardetection_amd import pillar_ops, synth
from lidardetection_amd.pcdet.models.backbones_2d import map_to_bev
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone, vfe
from lidardetection_amd.pcdet.models.model_utils import model_nms_utils
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.voxelizer import BatchVoxelizer
from oracle import c_oracle, spconv_oracle as so

pytestmark = pytest.mark.gpu


def test_pillar_modules_vs_reference_golden(dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "pp_modules.npz"))
    cfg = AttrDict(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=[64])
    m = vfe.PillarVFE(cfg, 4, [float(v) for v in g["voxel_size"]], [float(v) for v in g["pc_range"]]).to(dev)
    with torch.no_grad():
        m.pfn_layers[0].linear.weight.copy_(torch.from_numpy(g["pfn_weight"]))
        n = m.pfn_layers[0].norm
        n.weight.copy_(torch.from_numpy(g["bn_gamma"])); n.bias.copy_(torch.from_numpy(g["bn_beta"]))
        n.running_mean.copy_(torch.from_numpy(g["bn_mean"])); n.running_var.copy_(torch.from_numpy(g["bn_var"]))
    bd = {"voxels": torch.from_numpy(g["voxels"]).to(dev), "voxel_num_points": torch.from_numpy(g["num_points"]).float().to(dev),
          "voxel_coords": torch.from_numpy(g["coords"]).float().to(dev), "batch_size": 2}
    for mode in ("eval", "train_path"):     # fused HIP kernel, then the stock-torch path (BN kept in eval to compare)
        m.eval()
        if mode == "train_path":
            bd["voxels"] = bd["voxels"].clone().requires_grad_(True)     # forces the torch path
        out = m(dict(bd))["pillar_features"]
        np.testing.assert_allclose(out.detach().cpu().numpy(), g["pillar_features"], rtol=0, atol=1e-4)
    bd["voxels"] = bd["voxels"].detach()
    bd = m(bd)
    sc = map_to_bev.PointPillarScatter(AttrDict(NUM_BEV_FEATURES=64), grid_size=(48, 40, 1))
    canvas = sc(bd)["spatial_features"]
    ref = np.zeros(tuple(g["canvas_shape"]), np.float32)
    ref[tuple(g["canvas_nz_idx"])] = g["canvas_nz_val"]
    np.testing.assert_allclose(canvas.cpu().numpy(), ref, rtol=0, atol=1e-4)
    mv = vfe.MeanVFE(AttrDict(), 4)({"voxels": bd["voxels"], "voxel_num_points": bd["voxel_num_points"]})["voxel_features"]
    np.testing.assert_allclose(mv.cpu().numpy(), g["mean_features"], rtol=0, atol=1e-6)


def _oracle_sequential(seq, feats, idx, shape, B, dicts):
    """Replays a SparseSequential / SparseBasicBlock chain with the dense-conv oracle in float64."""
    from lidardetection_amd import spconv
    for mod in (seq._modules.values() if isinstance(seq, spconv.SparseSequential) else [seq]):
        if isinstance(mod, spconv.SparseSequential):
            feats, idx, shape = _oracle_sequential(mod, feats, idx, shape, B, dicts)
        elif isinstance(mod, spconv_backbone.SparseBasicBlock):
            ident = feats
            f1, idx, shape = _oracle_sequential(mod.conv1, feats, idx, shape, B, dicts)
            f1 = torch.relu(_bn(mod.bn1, f1))
            f2, idx, shape = _oracle_sequential(mod.conv2, f1, idx, shape, B, dicts)
            feats = torch.relu(_bn(mod.bn2, f2) + ident)
        elif isinstance(mod, spconv.SparseConvolution):
            w = mod.weight.detach().cpu()
            b = mod.bias.detach().cpu() if mod.bias is not None else None
            if mod.subm:
                feats = so.conv_features(feats, idx, B, shape, w, b, mod.kernel_size, [1, 1, 1], [0, 0, 0], True, idx)
            else:
                _, outs = so.rulebook(idx, shape, mod.kernel_size, mod.stride, mod.padding, False)
                outs = np.array(outs, np.int64)
                feats = so.conv_features(feats, idx, B, shape, w, b, mod.kernel_size, mod.stride, mod.padding, False, outs)
                idx, shape = outs, so.out_shape(shape, mod.kernel_size, mod.stride, mod.padding)
        elif isinstance(mod, torch.nn.BatchNorm1d):
            feats = _bn(mod, feats)
        elif isinstance(mod, torch.nn.ReLU):
            feats = torch.relu(feats)
        else:
            raise NotImplementedError(type(mod))
    return feats, idx, shape


def _bn(bn, x):
    d = lambda t: t.detach().cpu().double()
    return (x - d(bn.running_mean)) / torch.sqrt(d(bn.running_var) + bn.eps) * d(bn.weight) + d(bn.bias)


@pytest.mark.parametrize("res", [False, True])
def test_voxel_backbone8x_small_grid(dev, res):
    grid = [48, 40, 24]                                   # nx, ny, nz -> sparse_shape [25, 40, 48]
    B = 2
    r = np.random.default_rng(5)
    cells = 25 * 40 * 48
    pick = np.concatenate([r.choice(cells, 2500, replace=False) + b * cells for b in range(B)])
    b_, rem = np.divmod(pick, cells)
    z, rem = np.divmod(rem, 40 * 48)
    y, x = np.divmod(rem, 48)
    idx = np.stack([b_, z, y, x], 1).astype(np.int32)
    feats = r.standard_normal((len(idx), 4)).astype(np.float32)
    torch.manual_seed(3)
    cls = spconv_backbone.VoxelResBackBone8x if res else spconv_backbone.VoxelBackBone8x
    m = cls(AttrDict(), 4, grid).to(dev).eval()
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.8, 1.2); mod.bias.uniform_(-0.1, 0.1)
        bd = m({"voxel_features": torch.from_numpy(feats).to(dev), "voxel_coords": torch.from_numpy(idx).float().to(dev), "batch_size": B})
    out = bd["encoded_spconv_tensor"]
    f, i, s = torch.from_numpy(feats).double(), idx.astype(np.int64), m.sparse_shape
    for name in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
        f, i, s = _oracle_sequential(getattr(m, name), f, i, s, B, None)
    assert out.spatial_shape == list(s)
    got = {tuple(int(v) for v in c): row for c, row in zip(out.indices.cpu().numpy(), out.features.cpu().double().numpy())}
    assert set(got) == {tuple(int(v) for v in c) for c in i}
    ref = np.stack([got[tuple(int(v) for v in c)] for c in i])
    scale = max(1.0, float(f.abs().max()))
    np.testing.assert_allclose(ref, f.numpy(), rtol=0, atol=2e-4 * scale)
    hc = map_to_bev.HeightCompression(AttrDict(NUM_BEV_FEATURES=256))(bd)["spatial_features"]
    assert hc.shape == (B, 128 * s[0], s[1], s[2])
    assert set(bd["multi_scale_3d_features"]) == {"x_conv1", "x_conv2", "x_conv3", "x_conv4"}


def test_second_front_end_on_kitti_shapes(dev):
    """SECOND-KITTI shapes end to end: HIP voxelise (P=5) -> MeanVFE -> VoxelBackBone8x -> HeightCompression."""
    frames = [synth.cloud_ring(2000), synth.cloud_ring(2001)]
    vz = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000)
    o = vz.voxelize_frames(frames, device=dev)
    bd = {"voxels": o["voxels"], "voxel_num_points": o["voxel_num_points"], "voxel_coords": o["voxel_coords"], "batch_size": 2}
    bd = vfe.MeanVFE(AttrDict(), 4)(bd)
    m = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
    with torch.no_grad():
        bd = m(bd)
        bd = map_to_bev.HeightCompression(AttrDict(NUM_BEV_FEATURES=256))(bd)
    assert bd["spatial_features"].shape == (2, 256, 200, 176)
    assert bd["encoded_spconv_tensor"].spatial_shape == [2, 200, 176]
    assert torch.isfinite(bd["spatial_features"]).all()
    # the dense map is exactly the sparse rows scattered (height_compression.py:21-24): rebuild it from the rows
    t = bd["encoded_spconv_tensor"]
    dense = torch.zeros((2, 128, 2, 200, 176), device=dev)
    i = t.indices.long()
    dense[i[:, 0], :, i[:, 1], i[:, 2], i[:, 3]] = t.features
    assert torch.equal(bd["spatial_features"], dense.view(2, 256, 200, 176))


def test_second_backbone_full_grid_frame_vs_sparse_oracle(dev):
    """ONE full-size SECOND-KITTI frame (41 x 1600 x 1408 grid, ~16 k voxels) through VoxelBackBone8x, BatchNorm statistics
    perturbed: every tap (x_conv1..4, encoded tensor) against the sparse fp64 oracle (oracle/spconv_sparse_oracle.py: binary search
    over the active sites + one float64 matmul per kernel offset) — the dense conv3d oracle cannot hold this grid.  Active sites
    bit-exact as sets, features within 1e-4 of the feature scale (reference: spconv_backbone.py:76-163)."""
    from test_gpu_configs import _check_backbone_full_grid
    o = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000).voxelize_frames([synth.cloud_ring(2007)], device=dev)
    feats = pillar_ops.mean_vfe(o["voxels"], o["voxel_num_points"])
    torch.manual_seed(5)
    for cls in (spconv_backbone.VoxelBackBone8x, spconv_backbone.VoxelResBackBone8x):
        m = cls(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, torch.nn.BatchNorm1d):
                    mod.running_mean.uniform_(-0.2, 0.2); mod.running_var.uniform_(0.5, 1.5)
                    mod.weight.uniform_(0.8, 1.2); mod.bias.uniform_(-0.1, 0.1)
        errs = _check_backbone_full_grid(m, feats, o["voxel_coords"], dev)
        assert set(errs) == {"conv1", "conv2", "conv3", "conv4", "conv_out"}
        print(cls.__name__, "full grid, max err / scale:", {k: f"{v:.1e}" for k, v in errs.items()})


def test_model_nms_utils(dev):
    boxes, scores = synth.boxes_nms(seed=3020, objects=200, copies=8)
    cfg = AttrDict(NMS_TYPE="nms_gpu", NMS_THRESH=0.1, NMS_PRE_MAXSIZE=1024, NMS_POST_MAXSIZE=100, MULTI_CLASSES_NMS=False)
    tb, ts = torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev)
    sel, sc = model_nms_utils.class_agnostic_nms(ts, tb, cfg, score_thresh=0.3)
    # oracle replay of class_agnostic_nms
    mask = scores >= 0.3
    bs, ss = boxes[mask], scores[mask]
    top = np.argsort(-ss, kind="stable")[:1024]
    keep = c_oracle.nms(bs[top], ss[top], 0.1)[:100]
    exp = np.nonzero(mask)[0][top[keep]]
    assert sel.cpu().tolist() == exp.tolist() and np.allclose(sc.cpu().numpy(), scores[exp])
    # multi_classes_nms (model_nms_utils.py:28-65) replayed class by class against the oracle: per class the score mask, the
    # top NMS_PRE_MAXSIZE by score, the rotated NMS keep list, its first NMS_POST_MAXSIZE entries — concatenated in class order
    s3 = np.stack([scores, scores[::-1] * np.float32(0.9), np.roll(scores, 777) * np.float32(0.05)], 1).astype(np.float32)
    ps, pl, pb = model_nms_utils.multi_classes_nms(torch.from_numpy(s3).to(dev), tb, cfg, score_thresh=0.2)
    es, el, eb = [], [], []
    for k in range(s3.shape[1]):
        col = s3[:, k]
        msk = col >= np.float32(0.2)
        ids = np.nonzero(msk)[0]
        top = np.argsort(-col[msk], kind="stable")[:1024]
        kept = c_oracle.nms(boxes[ids][top], col[ids][top], 0.1)[:100] if len(top) else np.zeros(0, np.int64)
        chosen = ids[top[kept]] if len(top) else ids[:0]
        es.append(col[chosen]); el.append(np.full(len(chosen), k, np.int64)); eb.append(boxes[chosen])
    assert len(es[2]) == 0 and len(es[0]) == 100 and len(es[1]) > 0          # one class with nothing above the threshold
    assert np.array_equal(pl.cpu().numpy(), np.concatenate(el))
    assert np.array_equal(ps.cpu().numpy(), np.concatenate(es)) and np.array_equal(pb.cpu().numpy(), np.concatenate(eb))


def test_second_kitti_pipeline_runs_and_matches_unfused_paths(dev):
    """SECOND-KITTI end to end (voxelise -> MeanVFE -> sparse backbone -> dense -> folded BEV backbone -> HIP post-processing ->
    NMS) on two ring clouds: the fused inference stack must agree with the plain module / torch-op paths stage by stage."""
    from lidardetection_amd.second import SECONDKitti
    B = 2
    frames = [synth.cloud_ring(2000 + f) for f in range(B)]
    sizes = [len(f) for f in frames]
    pts = torch.from_numpy(np.concatenate(frames, 0)).to(dev)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    m = SECONDKitti(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(2)
    with torch.no_grad():
        feats, coords = m.voxelize_vfe(pts, offs)
        canvas = m.sparse_backbone(feats, coords)                      # fused conv+BN+ReLU, mask-ordered GEMM
    with torch.enable_grad():
        bd = m.backbone3d({"voxel_features": feats, "voxel_coords": coords, "batch_size": B})   # module sequence
    ref = bd["encoded_spconv_tensor"].dense().detach()
    ref = ref.view(B, -1, ref.shape[3], ref.shape[4])
    assert canvas.shape == (B, 256, 200, 176)
    scale = float(ref.abs().max())
    assert float((canvas - ref).abs().max()) <= 1e-4 * max(scale, 1.0)
    with torch.no_grad():
        (head,) = m.backbone_head(canvas)
        assert head.shape == (B, 200, 176, 6 * (3 + 7 + 2))
        fused = m.post_process(head)
        plain = m.post_process(*m.split_heads(head))
        full = m(pts, offs)
    for x, y, z in zip(fused, plain, full):
        assert torch.equal(x, y) and torch.equal(x, z)
    assert int(fused[3].min()) >= 0 and torch.isfinite(fused[0]).all()


def test_pipelined_stage_runner_equals_prebuilt_pass(dev):
    """spconv.run_stages_pipelined (rulebooks of stage s+1 on a second stream under stage s's GEMMs) vs building every
    rulebook first and running the stages one after another: same tables, same sums — bit-identical taps, repeatedly
    (the second and third pass reuse the streams and catch a missing cross-stream dependency)."""
    from lidardetection_amd import spconv, synth
    from lidardetection_amd.voxelizer import BatchVoxelizer
    frames = [synth.cloud_ring(2000 + f) for f in range(3)]
    o = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000).voxelize_frames(frames, device=dev)
    feats = pillar_ops.mean_vfe(o["voxels"], o["voxel_num_points"])
    torch.manual_seed(11)
    m = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
    stages = [getattr(m, n) for n in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out")]
    with torch.no_grad():
        x = spconv.SparseConvTensor(feats, o["voxel_coords"].int(), m.sparse_shape, 3)
        spconv.prebuild_rulebooks(stages, x.indices.contiguous(), x.spatial_shape, x.batch_size, x.indice_dict)
        want = []
        for st in stages:
            x = st(x)
            want.append(x)
        for _ in range(3):
            x = spconv.SparseConvTensor(feats, o["voxel_coords"].int(), m.sparse_shape, 3)
            got = spconv.run_stages_pipelined(stages, x)
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert a.spatial_shape == b.spatial_shape and torch.equal(a.indices, b.indices)
                assert torch.equal(a.features, b.features)


def test_dense_gemm_with_strided_bias_relu_epilogue_matches_fp64(dev):
    """lidar_dense_gemm_bias_act (csrc/dense_gemm.hip; the stride-1 deblock of BaseBEVBackbone, base_bev_backbone.py:58-77,103):
    D[:, off:off+N] = relu(A @ W + b) written at the concatenated map's row pitch, the other channels untouched; plus the plain
    (M, N) form with and without bias the other backbone GEMMs use.  fp32 within 1e-4 of the fp64 result."""
    from lidardetection_amd import bev_backbone
    torch.manual_seed(5)
    B, K, N, CT, h, w, off = 3, 64, 128, 384, 20, 24, 128
    x = torch.randn(B, K, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wkn = (torch.randn(K, N, device=dev) * 0.2).contiguous()
    bias = torch.randn(N, device=dev)
    cat = torch.full((B, CT, h, w), 7.0, device=dev).contiguous(memory_format=torch.channels_last)
    rows = x.permute(0, 2, 3, 1).reshape(-1, K)
    want = torch.relu(rows.double() @ wkn.double() + bias.double())
    if not bev_backbone.gemm_bias_act_into_(x, wkn, bias, cat, off):
        pytest.skip("hipBLASLt path not available in this process")
    got = cat.permute(0, 2, 3, 1).reshape(-1, CT)
    assert float((got[:, off:off + N].double() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    assert bool((got[:, :off] == 7.0).all()) and bool((got[:, off + N:] == 7.0).all())          # neighbours of the slice untouched
    for b in (None, bias):
        y = bev_backbone.rows_gemm(rows.contiguous(), wkn, b)
        ref = rows.double() @ wkn.double() + (0 if b is None else b.double())
        assert float((y.double() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))
    with pytest.raises(Exception):
        bev_backbone.gemm_bias_act_into_(x, wkn, bias, cat, CT - 8)                              # slice past the map's channels
