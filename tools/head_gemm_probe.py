"""Head (384 -> 72, 1x1) as MIOpen conv vs a plain GEMM (torch.addmm -> hipBLASLt/rocBLAS) on the NHWC map."""
import time
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True


def run(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


B, C, H, W, N = 16, 384, 248, 216, 72
x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = (torch.randn(N, C, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.randn(N, device=dev)
x2 = x.permute(0, 2, 3, 1).reshape(-1, C)
assert x2.data_ptr() == x.data_ptr()
wt = w.view(N, C).t().contiguous()
with torch.no_grad():
    ref = F.conv2d(x, w, b).permute(0, 2, 3, 1).reshape(-1, N)
    got = torch.addmm(b, x2, wt)
    ref64 = (x2[:4096].double() @ wt.double() + b.double())
    print("addmm vs conv maxdiff %.2e; conv vs f64 %.2e; addmm vs f64 %.2e" % (
        (got - ref).abs().max().item(), (ref[:4096].double() - ref64).abs().max().item(),
        (got[:4096].double() - ref64).abs().max().item()))
    print("conv2d 1x1 + bias   %.3f ms" % run(lambda: F.conv2d(x, w, b)))
    print("addmm (N=72)        %.3f ms" % run(lambda: torch.addmm(b, x2, wt)))
    print("mm    (N=72)        %.3f ms" % run(lambda: torch.mm(x2, wt)))
    for Np in (80, 96, 128):
        wp = torch.zeros(C, Np, device=dev); wp[:, :N] = wt
        print("mm    (N=%3d)       %.3f ms" % (Np, run(lambda: torch.mm(x2, wp))))
    # deblock 3 as a GEMM: (B*62*54, 256) @ (256, 4*4*128)
    a = torch.randn(B * 62 * 54, 256, device=dev); wd = torch.randn(256, 2048, device=dev) * 0.05
    print("deblock3 as mm      %.3f ms" % run(lambda: torch.mm(a, wd)))
    a = torch.randn(B * 124 * 108, 128, device=dev); wd = torch.randn(128, 512, device=dev) * 0.05
    print("deblock2 as mm      %.3f ms" % run(lambda: torch.mm(a, wd)))
    a = torch.randn(B * 248 * 216, 64, device=dev); wd = torch.randn(64, 128, device=dev) * 0.05
    print("deblock1 as mm      %.3f ms" % run(lambda: torch.mm(a, wd)))
