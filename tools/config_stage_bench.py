"""Per-stage GPU time of the PV-RCNN-KITTI (bs 8) and SECOND-MultiHead-NuScenes (bs 4) forwards (HIP events, medians)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth

dev = torch.device("cuda:0")


def gpu_time(fn, n=7):
    for _ in range(2):
        r = fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in evs:
        a.record(); r = fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])), r


def batch(frames):
    sizes = [len(f) for f in frames]
    pts = torch.from_numpy(np.concatenate(frames, 0)).to(dev)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    return pts, offs, sizes


which = sys.argv[1:] or ["pvrcnn", "multihead"]
with torch.no_grad():
    if "pvrcnn" in which:
        from lidardetection_amd.pvrcnn import PVRCNNKitti
        B = 8
        pts, offs, sizes = batch([synth.cloud_ring(2000 + f) for f in range(B)])
        m = PVRCNNKitti(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(0)
        m(pts, offs, sizes)
        t0, (ms3d, bev, head) = gpu_time(lambda: m.trunk(pts, offs))
        t1, (rois, rs, rl, _, _) = gpu_time(lambda: m.proposals(head))
        t2, kp = gpu_time(lambda: m.keypoints(pts, offs, sizes))
        t3, (before, fused) = gpu_time(lambda: m.set_abstraction(pts, sizes, kp, ms3d, bev))
        t4, ps = gpu_time(lambda: torch.sigmoid(m.point_cls_layers(before)).max(dim=-1)[0])
        t5, (rc, bx) = gpu_time(lambda: m.roi_head(rois, kp, fused, ps))
        t6, _ = gpu_time(lambda: m.final_nms(rc, bx, rl))
        tt, _ = gpu_time(lambda: m(pts, offs, sizes))
        print(f"PV-RCNN bs {B}: total {tt:.2f} ms = {B / tt * 1e3:.0f} frames/s | trunk {t0:.2f} proposals {t1:.2f} FPS keypoints {t2:.2f} "
              f"set abstraction {t3:.2f} point head {t4:.2f} roi head {t5:.2f} final nms {t6:.2f}", flush=True)
        # inside the set abstraction / roi head: ball query + group + MLP of one scale
        from lidardetection_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as pn
        new_xyz = kp.reshape(-1, 3).contiguous(); new_cnt = torch.full((B,), 2048, dtype=torch.int32, device=dev)
        xyz_cnt = torch.tensor(sizes, dtype=torch.int32, device=dev)
        xyz = pts[:, :3].contiguous(); f1 = pts[:, 3:].contiguous()
        tb, (idx, _) = gpu_time(lambda: pn.ball_query(0.4, 16, xyz, xyz_cnt, new_xyz, new_cnt))
        tg, g = gpu_time(lambda: pn.grouping_operation(f1, xyz_cnt, idx, new_cnt))
        print(f"  raw points scale 0: ball query {tb:.3f} group {tg:.3f} ms", flush=True)
        from lidardetection_amd.pvrcnn import roi_grid_points
        grid = roi_grid_points(rois.reshape(-1, 7), 6).reshape(-1, 3).contiguous()
        gc = torch.full((B,), 100 * 216, dtype=torch.int32, device=dev); kc = torch.full((B,), 2048, dtype=torch.int32, device=dev)
        kx = kp.reshape(-1, 3).contiguous()
        tb, (idx, _) = gpu_time(lambda: pn.ball_query(0.8, 16, kx, kc, grid, gc))
        tg, g = gpu_time(lambda: pn.grouping_operation(fused.contiguous(), kc, idx, gc))
        mlp = m.roi_grid_pool_layer.mlps[0]
        gg = torch.cat([torch.zeros(g.shape[0], 3, 16, device=dev), g], 1)
        tm, _ = gpu_time(lambda: mlp(gg.permute(1, 0, 2).unsqueeze(0)).amax(-1))
        print(f"  roi grid scale 0, reference layout (M, C, ns): ball query {tb:.3f} group {tg:.3f} MLP+max {tm:.3f} ms (M = {grid.shape[0]})", flush=True)
        tf, _ = gpu_time(lambda: m.roi_grid_pool_layer(xyz=kx, xyz_batch_cnt=kc, new_xyz=grid, new_xyz_batch_cnt=gc, features=fused.contiguous()))
        print(f"  roi grid pooling as the forward runs it (row-major groups + folded GEMM chain, both scales): {tf:.3f} ms", flush=True)
        del m
        torch.cuda.empty_cache()
    if "multihead" in which:
        from lidardetection_amd.second_multihead import SECONDMultiHeadNuScenes
        B = 4
        pts, offs, sizes = batch([synth.cloud_nus(4000 + f) for f in range(B)])
        m = SECONDMultiHeadNuScenes(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(0)
        m(pts, offs)
        t0, (feats, coords) = gpu_time(lambda: m.voxelize_vfe(pts, offs))
        t1, canvas = gpu_time(lambda: m.sparse_backbone(feats, coords))
        t2, sp2 = gpu_time(lambda: m.bev_features(canvas))
        t3, ho = gpu_time(lambda: m.heads(sp2))
        t4, (sc, bx, cn, lab) = gpu_time(lambda: m.candidates(ho))
        t5, _ = gpu_time(lambda: m.batched_class_nms(sc, bx, cn))
        tt, _ = gpu_time(lambda: m(pts, offs))
        print(f"SECOND-MultiHead bs {B}: total {tt:.2f} ms = {B / tt * 1e3:.0f} frames/s | voxelise+VFE {t0:.2f} ({feats.shape[0]} voxels) "
              f"sparse backbone {t1:.2f} BEV backbone {t2:.2f} heads {t3:.2f} candidates {t4:.2f} class NMS {t5:.2f}", flush=True)
