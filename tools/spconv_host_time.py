"""How long does the HOST need to enqueue one sparse-backbone forward (no read-backs inside)?  -> host enqueue ms vs GPU ms."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone, vfe
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.spconv import ops
from lidardetection_amd.voxelizer import BatchVoxelizer
dev = torch.device("cuda:0"); B = 16
frames = [synth.cloud_ring(2000 + f) for f in range(B)]
o = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000).voxelize_frames(frames, device=dev)
bd = {"voxels": o["voxels"], "voxel_num_points": o["voxel_num_points"], "voxel_coords": o["voxel_coords"], "batch_size": B}
bd = vfe.MeanVFE(AttrDict(), 4)(bd)
name = sys.argv[1] if len(sys.argv) > 1 else "VoxelBackBone8x"
m = getattr(spconv_backbone, name)(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
marks = []
orig = ops.resolve_speculation
ops.resolve_speculation = lambda spec: (marks.append(time.perf_counter()), orig(spec))[1]
with torch.no_grad():
    for _ in range(5): m(dict(bd))
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(30):
        t0 = time.perf_counter()
        m(dict(bd))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        host.append(marks[-1] - t0); total.append(t1 - t0)
print(f"{name}: host enqueue {1e3 * sorted(host)[len(host) // 2]:.2f} ms, forward {1e3 * sorted(total)[len(total) // 2]:.2f} ms (medians of 30)")
