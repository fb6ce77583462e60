"""Runs the folded dense backbone + head a few times, idles, then once more (for tools/ktrace_last.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd.pointpillar import PointPillarKITTI
dev = torch.device("cuda:0")
m = PointPillarKITTI(batch_size=16, device=dev).randomize_for_bench(0)
x = torch.randn(16, 64, 496, 432, device=dev)
x[:, :, ::3] = 0
x = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(4):
        m.backbone_head(x)
    torch.cuda.synchronize()
    time.sleep(1.0)
    m.backbone_head(x)
    torch.cuda.synchronize()
