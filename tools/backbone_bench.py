"""Dense 2D backbone + head of PointPillar-KITTI (bs 16, fp32): stock modules vs the folded-BN fast path
(lidardetection_amd/bev_backbone.py), plus a probe of torch's fused MIOpen conv+bias+relu op."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd.pointpillar import PointPillarKITTI
dev = torch.device("cuda:0")
m = PointPillarKITTI(batch_size=16, device=dev).randomize_for_bench(0)
x = torch.randn(16, 64, 496, 432, device=dev)
x[:, :, ::3] = 0
x = x.contiguous(memory_format=torch.channels_last)


def run(fn, n=10):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print("stock modules (channels_last)  %.2f ms" % run(lambda: m.backbone_head_stock(x)), flush=True)
print("folded + HIP epilogue          %.2f ms" % run(lambda: m.backbone_head(x)), flush=True)
bev = m._bev
print("  features only                %.2f ms" % run(lambda: bev.features(x)), flush=True)
with torch.no_grad():
    a, b = m.split_heads(m.backbone_head(x)[0]), m.backbone_head_stock(x)
print("max abs diff cls %.2e box %.2e dir %.2e" % tuple((p - q).abs().max().item() for p, q in zip(a, b)))
# probe: MIOpen fusion plan conv+bias+relu through torch
try:
    w, bb, stride, pad = bev.stages[0][0][1][:4]
    y0 = F.conv2d(x, bev.stages[0][0][0][0], None, 2, 1)
    from lidardetection_amd.bev_backbone import bias_act_
    bias_act_(y0, bev.stages[0][0][0][1])
    ref = torch.relu(F.conv2d(y0, w, bb, stride, pad))
    got = torch.miopen_convolution_relu(y0, w, bb, list(stride), [pad, pad], [1, 1], 1)
    print("miopen_convolution_relu diff %.2e" % (got - ref).abs().max().item())
    print("  conv + HIP epilogue  %.3f ms" % run(lambda: bias_act_(F.conv2d(y0, w, None, stride, pad), bb), 20))
    print("  miopen conv_relu     %.3f ms" % run(lambda: torch.miopen_convolution_relu(y0, w, bb, list(stride), [pad, pad], [1, 1], 1), 20))
    print("  conv only            %.3f ms" % run(lambda: F.conv2d(y0, w, None, stride, pad), 20))
except Exception as e:  # noqa: BLE001
    print("miopen_convolution_relu probe failed:", repr(e)[:300])
