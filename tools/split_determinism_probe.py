"""Is FoldedBEVBackbone.merged bit-reproducible from call to call, as one batch and as two part-batches on two streams?
(MIOpen's global-K-split convolution kernels accumulate with atomics; which kernels it picks depends on the batch size.)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.backends.cudnn.benchmark = True
from lidardetection_amd import bev_backbone
from lidardetection_amd.pointpillar import PointPillarKITTI
from lidardetection_amd.second import SECONDKitti
dev = torch.device("cuda:0")
B = 16
for name, model, shape in (("PointPillar", PointPillarKITTI(batch_size=B, max_voxels=16000, n_max=20000, device=dev).randomize_for_bench(0), (B, 64, 496, 432)),
                           ("SECOND", SECONDKitti(batch_size=B, n_max=20000, device=dev).randomize_for_bench(2), (B, 256, 200, 176))):
    torch.manual_seed(1)
    canvas = (torch.rand(shape, device=dev) * (torch.rand(shape[0], 1, shape[2], shape[3], device=dev) < 0.08)).contiguous(memory_format=torch.channels_last)
    bev = model._bev_folded()
    with torch.no_grad():
        for split in (1, 2):
            bev_backbone._SPLIT[0] = split
            ref = bev.merged(canvas).clone()
            diffs = []
            for _ in range(8):
                o = bev.merged(canvas)
                diffs.append(float((o - ref).abs().max()))
            torch.cuda.synchronize()
            print(f"{name} split={split}: max |diff| to the first call over 8 more calls: {max(diffs):.3e} (bit-identical: {max(diffs) == 0.0}), |out| max {float(ref.abs().max()):.2f}")
