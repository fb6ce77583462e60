"""The two assembled bs = 16 forwards that bench.py / tools/second_bench.py TIME, under the oracle.

One model instance, three successive DIFFERENT batches (uniform -> ring -> short / zero-padded / empty frames), the stages
called exactly as bench.py's timed loop calls them (voxelize -> vfe_scatter -> backbone_head -> post_process; SECOND:
voxelize_vfe -> sparse_backbone -> backbone_head -> post_process), i.e. THROUGH the resident voxel buffer, the `num_voxels_dev`
PFN fast path, the resident BEV canvas, the speculative sparse-conv capacities and the fused post-processing.  After every
step, on the buffers the model itself holds:
  * voxels / coords / counts / offsets   == sequential voxel oracle per frame (bit-exact), rows past the total all zero;
  * PFN rows                              vs the torch-CPU PillarVFE restatement (1e-4), MeanVFE 1e-5;
  * resident canvas                       == oracle PointPillarScatter of those rows (bit-exact), untouched cells exactly 0;
  * SECOND dense BEV map                  vs the fp64 sparse oracle replay of VoxelBackBone8x on the whole batch (1e-4 of scale);
  * boxes / scores / labels / counts      == the reference's control flow on the model's own head output: sigmoid -> max ->
    SCORE_THRESH mask -> top-k (NMS_PRE_MAXSIZE) -> ResidualCoder decode + direction bins -> rotated NMS (C oracle) -> first
    NMS_POST_MAXSIZE (pcdet/models/detectors/detector3d_template.py:169-275, model_nms_utils.py:6-25).
Top-k ties: an empty BEV region gives thousands of bit-equal scores, and the reference's torch.topk leaves the order among equal
scores unspecified.  The test checks the selection with the complete characterisation of a valid top-k (every score above the k-th
selected, the rest equal to it, no duplicates, sorted) AND against the rule this repo's selection follows (csrc/topk.hip: ties by
ascending anchor index = a stable CPU sort), then replays decode + NMS on that order."""
import numpy as np
import pytest
import torch

from lidardetection_amd import pillar_ops, synth
from oracle import c_oracle, pp_oracle

pytestmark = pytest.mark.gpu
B = 16


def _zero_padded(frame, n_total):
    return np.concatenate([frame, np.zeros((n_total - len(frame), frame.shape[1]), np.float32)], 0)


def _batches(pc_range):
    """three 16-frame batches of different character; `short` mixes truncated, zero-padded (collate_batch-style), tiny and
    EMPTY frames — every batch keeps n_max <= 20 000"""
    uni = [synth.cloud_uniform(1000 + f, pc_range=pc_range) for f in range(B)]
    ring = [synth.cloud_ring(2000 + f) for f in range(B)]
    short = []
    for f in range(B):
        src = synth.cloud_ring(2100 + f) if f % 2 else synth.cloud_uniform(1100 + f, pc_range=pc_range)
        n = [20000, 12345, 7000, 333, 0, 19968, 1, 15000][f % 8]
        fr = src[:min(n, len(src))]
        if f in (1, 6):
            fr = _zero_padded(fr, 18000)
        short.append(fr)
    return {"uniform": uni, "ring": ring, "short": short}


def _to_device(frames, dev):
    sizes = [len(f) for f in frames]
    pts = torch.from_numpy(np.concatenate(frames, 0)).to(dev)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    return pts, offs


def _check_voxels(vox, frames, vs, rng, P, maxv, tag):
    """the model's (resident) voxel buffers vs the sequential oracle, frame by frame; -> oracle (voxels, coords4, counts)"""
    offsets = vox["voxel_offsets"].cpu().numpy()
    exp = [c_oracle.voxelize(f, vs, rng, P, maxv) for f in frames]
    assert offsets.tolist() == np.concatenate([[0], np.cumsum([len(e[0]) for e in exp])]).tolist(), tag
    ev, ec, en = pp_oracle.collate(exp)
    total = int(offsets[-1])
    assert np.array_equal(vox["voxels"][:total].cpu().numpy().view(np.uint32), ev.view(np.uint32)), tag
    assert np.array_equal(vox["voxel_coords"][:total].cpu().numpy(), ec.astype(np.int32)), tag
    assert np.array_equal(vox["voxel_num_points"][:total].cpu().numpy(), en), tag
    assert not vox["voxels"][total:].any(), f"{tag}: resident buffer holds data past the produced rows"
    return ev, ec, en


def _check_post(m, head, out, tag):
    """final detections vs the reference's control flow on the model's own head output (see the module docstring)"""
    out_boxes, out_scores, out_labels, num = out
    cls, box, dirs = m.split_heads(head)
    scores_all, labels_all = torch.sigmoid(cls).max(dim=-1)                       # detector3d_template.py:205-230
    masked = torch.where(scores_all >= m.score_thresh, scores_all, scores_all.new_full((), -1.0))
    k = min(m.nms_pre, masked.shape[1])
    # the device's selection (recomputed exactly as post_process_fused does; the same kernels, deterministic)
    from lidardetection_amd import anchor_post
    got_masked, got_labels = anchor_post.anchor_scores(head, m.num_anchor_per_loc, m.num_class, m.score_thresh, cls_off=0)
    assert torch.equal(got_masked, masked) and torch.equal(got_labels.long(), labels_all), tag
    top_scores, top_idx = m.select_topk(got_masked, k)
    ties = 0
    for f in range(m.B):
        c = int((top_scores[f] >= m.score_thresh).sum())                 # valid entries (slots past them hold (-1, 0))
        assert c == min(k, int((masked[f] >= m.score_thresh).sum())), f"{tag} frame {f}: number of candidates"
        s, i = top_scores[f][:c], top_idx[f][:c].long()
        assert torch.equal(masked[f][i], s), f"{tag} frame {f}: reported scores are not the scores of the reported anchors"
        assert bool((s[:-1] >= s[1:]).all()), f"{tag} frame {f}: top-k not sorted"
        assert int(torch.unique(i).numel()) == c, f"{tag} frame {f}: duplicate anchors in the top-k"
        if c == k:
            kth = s[-1]
            assert int((masked[f] > kth).sum()) == int((s > kth).sum()), f"{tag} frame {f}: a score above the k-th was left out"
            ties += int((masked[f] == kth).sum()) - int((s == kth).sum())
    # the HIP selection is deterministic: (score desc, anchor index asc) — exactly a stable CPU sort of the candidates
    mc = masked.cpu().numpy()
    for f in range(m.B):
        valid = np.nonzero(mc[f] >= np.float32(m.score_thresh))[0]
        order = valid[np.lexsort((valid, -mc[f][valid].astype(np.float64)))][:k]
        assert np.array_equal(top_idx[f, :len(order)].cpu().numpy(), order), f"{tag} frame {f}: top-k order (ties by ascending index)"
    counts = (top_scores >= m.score_thresh).sum(dim=1)
    gi = top_idx.long().unsqueeze(-1)
    boxes = m.decode(torch.gather(box, 1, gi.expand(-1, -1, 7)), m.anchors[top_idx.long()],
                     torch.gather(dirs, 1, gi.expand(-1, -1, m.num_dir_bins))).contiguous()   # box_coder_utils.py:45-77 + dir bins
    bc, sc, lc = boxes.cpu().numpy(), top_scores.cpu().numpy(), labels_all.cpu().numpy()
    ic, cn = top_idx.cpu().numpy(), counts.cpu().numpy()
    ob, os_, ol, on = out_boxes.cpu().numpy(), out_scores.cpu().numpy(), out_labels.cpu().numpy(), num.cpu().numpy()
    for f in range(m.B):
        c = int(cn[f])
        keep = c_oracle.nms_sorted(np.ascontiguousarray(bc[f, :c]), m.nms_thresh)[:m.nms_post] if c else np.zeros(0, np.int64)
        assert int(on[f]) == len(keep), f"{tag} frame {f}: kept {int(on[f])}, oracle {len(keep)}"
        assert np.array_equal(ob[f, :len(keep)].view(np.uint32), bc[f][keep].view(np.uint32)), f"{tag} frame {f}: boxes"
        assert np.array_equal(os_[f, :len(keep)], sc[f][keep]), f"{tag} frame {f}: scores"
        assert np.array_equal(ol[f, :len(keep)], lc[f][ic[f][keep]] + 1), f"{tag} frame {f}: labels"
    return {"kept": on.tolist(), "candidates": cn.tolist(), "excluded_ties_at_kth": ties}


def test_pointpillar_kitti_bs16_timed_path_over_successive_batches(dev):
    from lidardetection_amd.pointpillar import PointPillarKITTI
    torch.manual_seed(0)
    m = PointPillarKITTI(batch_size=B, max_voxels=16000, n_max=20000, device=dev).randomize_for_bench(0)
    assert m.resident_voxels and m.channels_last and m.fold_bn           # the configuration bench.py times
    n = m.pfn_norm
    for tag, frames in _batches(synth.PP_RANGE).items():
        pts, offs = _to_device(frames, dev)
        hoffs = np.concatenate([[0], np.cumsum([len(f) for f in frames])]).tolist()   # as bench.py: the offsets on the host too
        with torch.no_grad():                                            # bench.py's timed loop, stage by stage
            vox = m.voxelize(pts, offs, hoffs)
            pmap = m.vfe_scatter(vox)                                    # a PillarMap: the folded backbone starts from the pillars
            (head,) = m.backbone_head(pmap)
            out = m.post_process(head)
            assert not torch.is_tensor(pmap) and m._bev_folded().sparse_first_ok()
            canvas = pmap.dense()                                        # the canvas everybody else would get (resident buffer)
            # the sparse first layer (neighbour table over all output pixels + mask-ordered implicit GEMM, csrc/pillar.hip) against
            # the stock convolution of that canvas: ZeroPad2d(1) + Conv2d(3x3, stride 2) + folded BatchNorm + ReLU
            bev = m._bev_folded()
            x1 = bev.first_layer_from_pillars(pmap)
            w0, b0, st0, pad0, _ = bev.stages[0][0][0]
            ref1 = torch.relu(torch.nn.functional.conv2d(canvas, w0, b0, st0, pad0))
            assert x1.shape == ref1.shape and x1.is_contiguous(memory_format=torch.channels_last)
            e1 = float((x1 - ref1).abs().max())
            assert e1 <= 1e-4 * max(1.0, float(ref1.abs().max())), (tag, e1)
            del ref1, x1
        ev, ec, en = _check_voxels(vox, frames, synth.PP_VOXEL, synth.PP_RANGE, 32, 16000, tag)
        total = len(ev)
        # PFN rows: the canvas holds them; recompute the kernel's rows with the model's own arguments (same kernel, same bits)
        w, s, t = m._pfn_folded()
        feat = pillar_ops.pillar_vfe(vox["voxels"], vox["voxel_num_points"], vox["voxel_coords"], w, s, t, m.voxel_size, m.pc_range,
                                     num_voxels_dev=vox["voxel_offsets"][B:B + 1])[:total]
        ref = pp_oracle.pillar_vfe(torch.from_numpy(ev), torch.from_numpy(en).float(), torch.from_numpy(ec).float(),
                                   m.pfn_linear.weight.detach().cpu(), n.weight.detach().cpu(), n.bias.detach().cpu(),
                                   n.running_mean.cpu(), n.running_var.cpu(), synth.PP_VOXEL, synth.PP_RANGE, eps=n.eps)
        err = float((feat.cpu() - ref).abs().max()) if total else 0.0
        assert err <= 1e-4, (tag, err)                                   # north_star tolerance for fp32 features
        # resident canvas == a fresh PointPillarScatter of those rows, bit for bit (pointpillar_scatter.py:14-37)
        want = pp_oracle.pillar_scatter(feat.cpu(), torch.from_numpy(ec).float(), B, m.nx, m.ny)
        assert canvas.shape == want.shape and canvas.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(canvas.cpu(), want), f"{tag}: resident canvas differs from the oracle scatter"
        del want
        info = _check_post(m, head, out, tag)
        print(f"[pp bs16 {tag}] rows {total} pfn err {err:.1e} kept/frame min {min(info['kept'])} max {max(info['kept'])} "
              f"candidates min {min(info['candidates'])} ties left out at the k-th score {info['excluded_ties_at_kth']}")
        assert max(info["kept"]) > 0
    # the whole forward in one call gives the same detections as the staged calls of the last batch
    # the voxeliser's other entry (device offsets only) gives the same bits as the host-offset entry used above.  (The whole forward
    # is NOT compared bit for bit between two calls: MIOpen's global-K-split convolution kernels accumulate with atomics, so the head
    # output moves in its last bit from call to call — 1.2e-7 on PointPillar, tools/split_determinism_probe.py — and every call is
    # checked against the oracle built from its own head output instead.)
    with torch.no_grad():
        keys = ("voxels", "voxel_coords", "voxel_num_points", "voxel_offsets")
        ref = {k: m._vox_out[k].clone() for k in keys}
        m.voxelizer(pts, offs, m.n_max, compact=True, out=m._vox_out, resident=m.resident_voxels)
        total = int(ref["voxel_offsets"][-1])
        for k in keys:
            n_rows = total if k != "voxel_offsets" else ref[k].numel()
            assert torch.equal(m._vox_out[k][:n_rows], ref[k][:n_rows]), f"device-offset entry: {k}"


def test_second_kitti_bs16_timed_path_over_successive_batches(dev):
    from lidardetection_amd.second import SECONDKitti
    from test_gpu_configs import _replay_sparse
    torch.manual_seed(0)
    m = SECONDKitti(batch_size=B, n_max=20000, device=dev).randomize_for_bench(2)
    bb = m.backbone3d
    for tag, frames in _batches(synth.SEC_RANGE).items():
        pts, offs = _to_device(frames, dev)
        hoffs = np.concatenate([[0], np.cumsum([len(f) for f in frames])]).tolist()
        with torch.no_grad():                                            # tools/second_bench.py's timed loop
            feats, coords = m.voxelize_vfe(pts, offs, hoffs)
            canvas = m.sparse_backbone(feats, coords)
            (head,) = m.backbone_head(canvas)
            out = m.post_process(head)
        ev, ec, en = _check_voxels(m._vox_out, frames, synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000, tag)
        total = len(ev)
        assert feats.shape[0] == total and torch.equal(coords.cpu(), torch.from_numpy(ec.astype(np.int32)))
        mean = ev.sum(1) / np.maximum(en, 1)[:, None].astype(np.float32)           # mean_vfe.py:14-31
        np.testing.assert_allclose(feats.cpu().numpy(), mean, rtol=0, atol=1e-5)
        # VoxelBackBone8x on the WHOLE batch, fp64, active sites only (spconv_backbone.py:119-163)
        f, idx, shape = feats.cpu().double().numpy(), ec.astype(np.int64), bb.sparse_shape
        for name in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
            f, idx, shape = _replay_sparse(getattr(bb, name), f, idx, shape)
        D, H, W = shape
        assert canvas.shape == (B, f.shape[1] * D, H, W)
        dense = np.zeros((B, f.shape[1], D, H, W), np.float32)                     # height_compression.py:21-25
        dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = f.astype(np.float32)
        dense = dense.reshape(B, -1, H, W)
        got = canvas.cpu().numpy()
        scale = max(1.0, float(np.abs(dense).max()))
        err = float(np.abs(got - dense).max()) / scale
        assert err <= 1e-4, (tag, err)
        occupied = np.zeros((B, D, H, W), bool)
        occupied[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = True
        empty_cells = ~np.repeat(occupied[:, None], f.shape[1], 1).reshape(B, -1, H, W)
        assert not got[empty_cells].any(), f"{tag}: inactive BEV cells must be exactly zero"
        info = _check_post(m, head, out, tag)
        print(f"[second bs16 {tag}] voxels {total} out sites {len(idx)} dense err/scale {err:.1e} kept/frame min {min(info['kept'])} "
              f"max {max(info['kept'])} ties left out at the k-th score {info['excluded_ties_at_kth']}")
    # the voxeliser's other entry (device offsets only) gives the same bits as the host-offset entry used above.  (The whole forward
    # is NOT compared bit for bit between two calls: MIOpen's global-K-split convolution kernels accumulate with atomics, so the head
    # output moves in its last bit from call to call — 1.2e-7 on PointPillar, tools/split_determinism_probe.py — and every call is
    # checked against the oracle built from its own head output instead.)
    with torch.no_grad():
        keys = ("voxels", "voxel_coords", "voxel_num_points", "voxel_offsets")
        ref = {k: m._vox_out[k].clone() for k in keys}
        m.voxelizer(pts, offs, m.n_max, compact=True, out=m._vox_out, resident=m.resident_voxels)
        total = int(ref["voxel_offsets"][-1])
        for k in keys:
            n_rows = total if k != "voxel_offsets" else ref[k].numel()
            assert torch.equal(m._vox_out[k][:n_rows], ref[k][:n_rows]), f"device-offset entry: {k}"


def test_pinned_feeder_batches_voxelise_bit_exactly(dev):
    """raw clouds staged in pinned memory and copied on a side stream while the previous batch is processed
    (lidardetection_amd/feeder.py; the reference: dataset.py:153-185 collate + models/__init__.py:16-22 synchronous .cuda()).
    Six successive batches of different sizes through TWO staging slots in bench.py's order (submit k+1, then get + voxelise k,
    with a long kernel in between so that copies and compute really overlap): every fed batch voxelises to the oracle's bits in
    the resident buffer, the host offsets it carries are the device offsets, and misuse is refused."""
    from lidardetection_amd import _lib
    from lidardetection_amd.feeder import PinnedPointFeeder
    from lidardetection_amd.voxelizer import BatchVoxelizer
    P, maxv = 32, 16000
    bt = _batches(synth.PP_RANGE)
    seq = [bt["uniform"][:5], bt["short"][:8], bt["ring"][:3], bt["short"][8:], bt["uniform"][5:9], [np.zeros((0, 4), np.float32)] * 2]
    feeder = PinnedPointFeeder(8 * 20000, 4, max_batch=8, device=dev, depth=2)
    vz = BatchVoxelizer(synth.PP_VOXEL, synth.PP_RANGE, P, maxv, 4)
    outs = {}
    busy = torch.empty(64 << 20, device=dev)
    feeder.submit(seq[0])
    for k, frames in enumerate(seq):
        if k + 1 < len(seq):
            feeder.submit(seq[k + 1])                    # overlaps with this batch's kernels
        fb = feeder.get()
        assert fb.batch == len(frames) and fb.host_offsets == [int(v) for v in np.concatenate([[0], np.cumsum([len(f) for f in frames])])]
        assert fb.offsets.cpu().tolist() == fb.host_offsets
        out = outs.setdefault(fb.batch, vz.alloc_outputs(fb.batch, dev))
        pts = fb.points if fb.points.shape[0] else torch.zeros((1, 4), device=dev)
        vox = vz(pts, fb.offsets, max(fb.n_max, 1), out=out, resident=False, host_offsets=fb.host_offsets)
        for _ in range(20):
            busy.mul_(1.0001)                            # keeps the compute stream busy while the next copy is in flight
        _check_voxels(vox, frames, synth.PP_VOXEL, synth.PP_RANGE, P, maxv, f"fed batch {k}")
    with pytest.raises(_lib.LidarHipError):
        feeder.get()                                     # nothing submitted
    feeder.submit(seq[0]); feeder.submit(seq[1])
    with pytest.raises(_lib.LidarHipError):
        feeder.submit(seq[2])                            # both slots un-fetched
    with pytest.raises(_lib.LidarHipError):
        PinnedPointFeeder(100, 4, max_batch=2, device=dev).submit(seq[0])      # does not fit
