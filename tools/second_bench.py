"""SECOND-KITTI (BASELINE.json configs[2]) forward + NMS, bs 16 on one MI355X: frames/s and per-stage GPU time."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.second import SECONDKitti

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
B = 16
frames = [synth.cloud_ring(2000 + f) for f in range(B)]
sizes = [len(f) for f in frames]
pts = torch.from_numpy(np.concatenate(frames, 0)).to(dev)
offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
m = SECONDKitti(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(0)


def gpu_time(fn, n=10):
    for _ in range(2): r = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): r = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n, r


with torch.no_grad():
    for _ in range(4): out = m(pts, offs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = m(pts, offs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"SECOND-KITTI bs {B}: {B / dt:.1f} frames/s ({dt * 1e3:.2f} ms/step), kept/frame {out[3].float().mean().item():.1f}")
    t1, (feats, coords) = gpu_time(lambda: m.voxelize_vfe(pts, offs))
    t2, canvas = gpu_time(lambda: m.sparse_backbone(feats, coords))
    t3, hb = gpu_time(lambda: m.backbone_head(canvas))
    t4, _ = gpu_time(lambda: m.post_process(*hb))
    print(f"[stages ms/batch] voxelize+MeanVFE {t1:.3f} ({feats.shape[0]} voxels)  sparse 3D backbone + dense {t2:.3f}  "
          f"BEV backbone + heads {t3:.3f}  post + NMS {t4:.3f}")
