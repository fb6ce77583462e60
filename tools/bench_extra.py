"""Side measurements that bench.py adds to its JSON line as `extra` (rank 0, N = 1, after the timed headline loop; they never
touch `value`): the other BASELINE.json configs end to end and the kernel-level figures north_star asks to see reported
(sparse-GEMM MFMA utilisation, NMS on the SURVEY §8d clustered box set, PFN).  Every number is a HIP-event time on the launch
stream; the rocprofv3 summaries that back them are under profiles/r02/."""
import time

import numpy as np
import torch

from lidardetection_amd import spconv, synth
from lidardetection_amd.ext import iou3d_nms_cuda
from lidardetection_amd.spconv import ops

FP32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense fp32 matrix peak


def _events(fn, n=10, warm=3):
    for _ in range(warm):
        r = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r


def _wall(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def _batch(frames, dev):
    sizes = [len(f) for f in frames]
    pts = torch.from_numpy(np.concatenate(frames, 0)).to(dev)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    return pts, offs, sizes


def second_kitti(dev, B=16):
    """BASELINE.json configs[2]: SECOND-KITTI forward + NMS, bs 16, ring clouds; plus the sparse stack's GEMM figures"""
    from lidardetection_amd.second import SECONDKitti
    pts, offs, sizes = _batch([synth.cloud_ring(2000 + f) for f in range(B)], dev)
    m = SECONDKitti(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(0)
    out = {}
    with torch.no_grad():
        dt = _wall(lambda: m(pts, offs))
        out["second_kitti"] = {"frames_per_s": B / dt, "ms_per_step": dt * 1e3, "batch": B}
        feats, coords = m.voxelize_vfe(pts, offs)
        ms, _ = _events(lambda: m.backbone3d({"voxel_features": feats, "voxel_coords": coords, "batch_size": B}))
        out["spconv_forward_ms"] = ms                                   # rulebooks + mask orders + fused GEMMs, VoxelBackBone8x
        # implicit-GEMM time of every sparse conv of the stack on its own table (mask order), useful FLOPs = 2 * pairs * Cin * Cout
        x = spconv.SparseConvTensor(feats, coords.int(), m.backbone3d.sparse_shape, B)
        tot_ms, tot_fl = 0.0, 0.0
        dev_pairs = []

        def walk(mod, x):
            nonlocal tot_ms, tot_fl
            for c in mod._modules.values():
                if isinstance(c, spconv.SparseSequential):
                    x = walk(c, x)
                elif isinstance(c, spconv.SparseConvolution):
                    y = c(x)
                    nbr = y.indice_dict[c.indice_key]["nbr"]
                    w = c.weight.reshape(-1, c.in_channels, c.out_channels).contiguous()
                    f = x.features.contiguous()
                    st = ops.mask_order(nbr) if ops.sorted_gemm_supported(w.shape[0], c.in_channels, c.out_channels) else None
                    pk = ops.pack_gemm_weights(w) if st is not None else None       # what forward_fused runs (csrc/sparse_conv.hip pk kernel)
                    t, _ = _events(lambda: ops.indice_conv_fused(f, nbr, w, None, None, True, st, pk), n=5, warm=2)
                    tot_ms += t
                    dev_pairs.append(int((nbr >= 0).sum()))
                    tot_fl += 2.0 * dev_pairs[-1] * c.in_channels * c.out_channels
                    x = y
                else:
                    x.features = c(x.features)
            return x
        for name in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
            x = walk(getattr(m.backbone3d, name), x)
        # the FLOP count is the ORACLE rulebook's (tests/golden/spconv_nk_second_kitti_bs16.json: per-layer n_k of this very
        # batch from the sparse fp64 oracle); the device tables must agree with it pair for pair
        fx, match = None, None
        try:
            import json, os
            with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "spconv_nk_second_kitti_bs16.json")) as fh:
                fx = json.load(fh)
            match = (B == 16) and [r["pairs"] for r in fx["layers"]] == dev_pairs
            if match:
                tot_fl = fx["gflop_useful_total"] * 1e9
        except Exception:
            pass
        tf = tot_fl / (tot_ms * 1e-3) / 1e12
        out["spconv_gemm"] = {"tflops_useful": tf, "frac_of_157.3": tf / FP32_MFMA_PEAK_TFLOPS, "ms": tot_ms,
                              "gflop_useful": tot_fl / 1e9, "pairs_equal_oracle_fixture": match,
                              "flop_source": "tests/golden/spconv_nk_second_kitti_bs16.json (oracle rulebook n_k)" if match else "device tables",
                              "note": "12 implicit GEMMs of VoxelBackBone8x, fp32 MFMA, mask-ordered rows"}
    return out


def nms_boxes(dev, B=16):
    """SURVEY §8d clustered set: 512 objects x 8 jittered copies = 4096 boxes per frame, 16 frames, FULL keep list"""
    bt = []
    for k in range(B):
        b, s = synth.boxes_nms(seed=3000 + k)
        bt.append(torch.from_numpy(b[np.argsort(-s, kind="stable")]))
    boxes = torch.stack(bt).to(dev)
    out = {}
    for thr in (0.01, 0.7):
        ms, (keep, num) = _events(lambda: iou3d_nms_cuda.nms_batch(boxes, None, thr), n=20)
        out[f"thr_{thr}"] = {"us_per_batch": ms * 1e3, "kept_per_frame": float(num.float().mean()),
                             "pair_tests_per_s": B * 4096 * 4095 / 2 / (ms * 1e-3)}
    return {"nms_boxes_nms_us": out["thr_0.01"]["us_per_batch"], "nms_boxes_nms": out}


def pvrcnn(dev, B=8):
    from lidardetection_amd.pvrcnn import PVRCNNKitti
    pts, offs, sizes = _batch([synth.cloud_ring(2000 + f) for f in range(B)], dev)
    m = PVRCNNKitti(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(0)
    with torch.no_grad():
        dt = _wall(lambda: m(pts, offs, sizes), n=5, warm=2)
        t_fps, _ = _events(lambda: m.keypoints(pts, offs, sizes), n=5, warm=1)
    return {"pvrcnn_kitti": {"frames_per_s": B / dt, "ms_per_step": dt * 1e3, "batch": B, "fps_keypoints_ms": t_fps}}


def multihead(dev, B=4):
    from lidardetection_amd.second_multihead import SECONDMultiHeadNuScenes
    pts, offs, sizes = _batch([synth.cloud_nus(4000 + f) for f in range(B)], dev)
    m = SECONDMultiHeadNuScenes(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(0)
    with torch.no_grad():
        dt = _wall(lambda: m(pts, offs), n=5, warm=2)
    return {"second_multihead_nuscenes": {"frames_per_s": B / dt, "ms_per_step": dt * 1e3, "batch": B,
                                           "note": "one rank's share of the 8-GPU DDP config (bs 4 per GPU), forward + per-class NMS"}}


def pp_kernels(model, pts, offs):
    """two kernels of the headline step by themselves, on the step's own data: the PFN launch, and NMS prep + mask (the greedy
    pass told to stop after its first survivor, so it adds ~nothing)"""
    from lidardetection_amd import anchor_post, pillar_ops
    with torch.no_grad():
        vox = model.voxelize(pts, offs)
        w, s, t = model._pfn_folded()
        total = vox["voxel_offsets"][model.B:model.B + 1]
        pfn_ms, _ = _events(lambda: pillar_ops.pillar_vfe(vox["voxels"], vox["voxel_num_points"], vox["voxel_coords"], w, s, t,
                                                          model.voxel_size, model.pc_range, num_voxels_dev=total), n=20)
        (head,) = model.backbone_head(model.vfe_scatter(vox))
        a = model.num_anchor_per_loc
        masked, _ = anchor_post.anchor_scores(head, a, model.num_class, model.score_thresh, cls_off=0)
        top_scores, top_idx = torch.topk(masked, model.nms_pre, dim=1)
        counts = (top_scores >= model.score_thresh).sum(dim=1).to(torch.int32)
        boxes = anchor_post.decode_topk(head, top_idx, model.anchors, a, box_off=a * model.num_class, dir_off=a * (model.num_class + 7),
                                        num_dir_bins=model.num_dir_bins, dir_offset=model.dir_offset, dir_limit_offset=model.dir_limit_offset)
        mask_ms, _ = _events(lambda: iou3d_nms_cuda.nms_batch(boxes, counts, model.nms_thresh, max_keep=1), n=20)
    rows = int(total.item())
    return {"pfn_us": pfn_ms * 1e3, "pfn_rows": rows, "nms_mask_us": mask_ms * 1e3,
            "nms_mask_note": f"prep + mask kernels on the step's own {model.B} x {model.nms_pre} candidates, thr {model.nms_thresh}"}


def pp_ring(model):
    """the voxeliser bracket inside full PointPillar steps on KITTI-like RING clouds (multi-point pillars, 8.9 k pillars per frame,
    a third of the points outside the grid) — the headline uses the uniform cloud; both output modes"""
    dev = model.anchors.device
    pts, offs, sizes = _batch([synth.cloud_ring(2000 + f) for f in range(model.B)], dev)
    hoffs = [int(v) for v in np.concatenate([[0], np.cumsum(sizes)])]
    out = {}
    keep = model.resident_voxels
    with torch.no_grad():
        for mode, resident in (("resident", True), ("contract", False)):
            model.resident_voxels = resident
            for _ in range(2):
                model(pts, offs, hoffs)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
            for a, b in evs:
                a.record()
                vox = model.voxelize(pts, offs, hoffs)
                b.record()
                model.post_process(*model.backbone_head(model.vfe_scatter(vox)))
            torch.cuda.synchronize()
            out[f"{mode}_us"] = float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e3
        out["rows"] = int(vox["voxel_offsets"][-1].item())
    model.resident_voxels = keep
    model(pts, offs, hoffs)
    return {"voxelize_ring_cloud": dict(out, note="HIP-event bracket of lidar_voxelize inside full steps, cloud_ring(2000..2015), 16 frames")}


def collect(dev):
    out = {}
    for fn in (second_kitti, nms_boxes, pvrcnn, multihead):
        try:
            out.update(fn(dev))
        except Exception as e:                          # a side measurement must never sink the headline line
            out[fn.__name__ + "_error"] = repr(e)[:200]
        torch.cuda.empty_cache()
    return out
