"""Times the three 3x3 stride-1 conv shapes of the PointPillar BEV backbone (bs 16, fp32) in NCHW vs NHWC under
MIOpen's find mode, to decide the per-block memory format."""
import os, sys, time
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True


def run(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for C, H, W in ((64, 248, 216), (128, 124, 108), (256, 62, 54)):
    x = torch.randn(16, C, H, W, device=dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    gf = 2 * 16 * H * W * C * C * 9 / 1e9
    with torch.no_grad():
        t1 = run(lambda: F.conv2d(x, w, None, 1, 1))
        xl, wl = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        t2 = run(lambda: F.conv2d(xl, wl, None, 1, 1))
        d = (F.conv2d(x, w, None, 1, 1) - F.conv2d(xl, wl, None, 1, 1)).abs().max().item()
    print(f"C={C:3d} {H}x{W}: NCHW {t1:.3f} ms ({gf / t1:.1f} TF)  NHWC {t2:.3f} ms ({gf / t2:.1f} TF)  maxdiff {d:.2e}", flush=True)
