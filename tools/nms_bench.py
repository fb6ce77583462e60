"""Micro-benchmark of the batched rotated NMS (16 frames x 4096 boxes). usage: python tools/nms_bench.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.ext import iou3d_nms_cuda
dev = torch.device("cuda:0")
B = 16
bt = []
for k in range(B):
    b, s = synth.boxes_nms(seed=3000 + k)
    bt.append(torch.from_numpy(b[np.argsort(-s, kind="stable")]))
boxes = torch.stack(bt).to(dev)
for thr in (0.01, 0.7):
    for _ in range(3):
        keep, num = iou3d_nms_cuda.nms_batch(boxes, None, thr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        keep, num = iou3d_nms_cuda.nms_batch(boxes, None, thr)
    e1.record(); torch.cuda.synchronize()
    print(f"thr {thr}: {e0.elapsed_time(e1)/20*1e3:.1f} us per batch of {B}x4096, kept/frame {num.float().mean().item():.0f}")
