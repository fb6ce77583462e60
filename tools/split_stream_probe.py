"""Would the BEV backbone + heads run faster as two half-batches on two streams (the HBM-bound epilogue / zero-fill / pixel-shuffle
passes of one half under the MFMA-bound convolutions of the other)?  Times FoldedBEVBackbone.merged on 16 frames against
2 x 8 frames on two streams, on the PointPillar canvas."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.backends.cudnn.benchmark = True
from lidardetection_amd.pointpillar import PointPillarKITTI
from lidardetection_amd.bev_backbone import FoldedBEVBackbone
dev = torch.device("cuda:0")
B = 16
m = PointPillarKITTI(batch_size=B, max_voxels=16000, n_max=20000, device=dev).randomize_for_bench(0)
canvas = (torch.rand(B, 64, 496, 432, device=dev) * (torch.rand(B, 1, 496, 432, device=dev) < 0.08)).contiguous(memory_format=torch.channels_last)
bev = m._bev_folded()
bev2 = [FoldedBEVBackbone(m.blocks, m.deblocks, [m.conv_cls, m.conv_box, m.conv_dir_cls]) for _ in range(2)]   # own concat buffers
streams = [torch.cuda.Stream(dev) for _ in range(2)]


def whole():
    return bev.merged(canvas)


def halves():
    cur = torch.cuda.current_stream(dev)
    outs = []
    for i, (s, b2) in enumerate(zip(streams, bev2)):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            outs.append(b2.merged(canvas[i * 8:(i + 1) * 8]))
    for s in streams:
        cur.wait_stream(s)
    return outs


def t(fn, n=10):
    with torch.no_grad():
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    a = whole(); b = halves()
    print("max |diff| whole vs halves:", float((a - torch.cat(b, 0)).abs().max()))
print(f"16 frames, one stream: {t(whole):.3f} ms")
print(f"2 x 8 frames, two streams: {t(halves):.3f} ms")
with torch.no_grad():
    print(f"8 frames alone, one stream: {t(lambda: bev2[0].merged(canvas[:8])):.3f} ms")
