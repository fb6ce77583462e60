"""PV-RCNN-KITTI bs 8 or SECOND-MultiHead-NuScenes bs 4: a few forwards for `rocprofv3 --kernel-trace --stats` (argument: pvrcnn | multihead)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "pvrcnn"
with torch.no_grad():
    if which == "pvrcnn":
        from lidardetection_amd.pvrcnn import PVRCNNKitti
        B = 8
        frames = [synth.cloud_ring(2000 + f) for f in range(B)]
        sizes = [len(f) for f in frames]
        pts = torch.from_numpy(np.concatenate(frames, 0)).to(dev)
        offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
        m = PVRCNNKitti(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(0)
        run = lambda: m(pts, offs, sizes)
    else:
        from lidardetection_amd.second_multihead import SECONDMultiHeadNuScenes
        B = 4
        frames = [synth.cloud_nus(4000 + f) for f in range(B)]
        sizes = [len(f) for f in frames]
        pts = torch.from_numpy(np.concatenate(frames, 0)).to(dev)
        offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
        m = SECONDMultiHeadNuScenes(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(0)
        run = lambda: m(pts, offs)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    for _ in range(10):
        run()
    torch.cuda.synchronize()
print("done", which)
