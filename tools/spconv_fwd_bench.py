"""SECOND-KITTI sparse backbone forward (bs 16, ring clouds): wall / GPU-event time per forward, no-grad (fused inference
path) vs grad-enabled (module sequence)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone, vfe
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.voxelizer import BatchVoxelizer
dev = torch.device("cuda:0"); B = 16
frames = [synth.cloud_ring(2000 + f) for f in range(B)]
o = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000).voxelize_frames(frames, device=dev)
bd = {"voxels": o["voxels"], "voxel_num_points": o["voxel_num_points"], "voxel_coords": o["voxel_coords"], "batch_size": B}
bd = vfe.MeanVFE(AttrDict(), 4)(bd)
for cls in (spconv_backbone.VoxelBackBone8x, spconv_backbone.VoxelResBackBone8x):
    m = cls(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
    for label, ctx in (("module sequence", torch.enable_grad), ("fused inference", torch.no_grad), ("one hipGraph", torch.no_grad)):
        m.graph_capacity = B * 16000 if label == "one hipGraph" else 0
        with ctx():
            for _ in range(3): m(dict(bd))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); w0 = time.perf_counter(); e0.record()
            for _ in range(10): m(dict(bd))
            e1.record(); torch.cuda.synchronize()
            print(f"{cls.__name__:20s} {label:16s}: wall {(time.perf_counter() - w0) / 10 * 1e3:.2f} ms, "
                  f"events {e0.elapsed_time(e1) / 10:.2f} ms per forward", flush=True)
