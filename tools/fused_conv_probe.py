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
    b = torch.randn(C, device=dev)
    gf = 2 * 16 * H * W * C * C * 9 / 1e9
    with torch.no_grad():
        ref = torch.relu(F.conv2d(x, w, b, 1, 1))
        t1 = run(lambda: F.conv2d(x, w, None, 1, 1))
        try:
            got = torch.miopen_convolution_relu(x, w, b, [1, 1], [1, 1], [1, 1], 1)
            d = (got - ref).abs().max().item()
            t2 = run(lambda: torch.miopen_convolution_relu(x, w, b, [1, 1], [1, 1], [1, 1], 1), 5)
        except Exception as e:
            d, t2 = float("nan"), float("nan"); print("fused failed", repr(e)[:200])
    print(f"C={C:3d} {H}x{W}: NCHW conv {t1:.3f} ms ({gf / t1:.1f} TF) | NCHW miopen_convolution_relu {t2:.3f} ms ({gf / t2:.1f} TF) maxdiff {d:.2e}", flush=True)
