"""One MIOpen setting per process (the environment must be in place before the library initialises): times the three stride-1
3x3 shapes of the PointPillar BEV backbone (bs 16, fp32) and names the kernels that ran (is there a
non-atomic solver that is as fast as the split-K one, and does any fp32 Winograd solver exist on gfx950?).
usage: python tools/conv_solver_probe.py LAYOUT(nhwc|nchw) DETERMINISTIC(0|1) [tag]    (env: MIOPEN_* as wanted)"""
import sys, time
import torch, torch.nn.functional as F
layout, det = sys.argv[1], sys.argv[2] == "1"
tag = sys.argv[3] if len(sys.argv) > 3 else ""
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.backends.cudnn.deterministic = det


def run(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


out = []
for C, H, W in ((64, 248, 216), (128, 124, 108), (256, 62, 54)):
    g = torch.Generator(device="cpu").manual_seed(C)
    x = torch.randn(16, C, H, W, generator=g).to(dev)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev)
    if layout == "nhwc":
        x, w = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    gf = 2 * 16 * H * W * C * C * 9 / 1e9
    try:
        with torch.no_grad():
            t = run(lambda: F.conv2d(x, w, None, 1, 1))
            y1 = F.conv2d(x, w, None, 1, 1); y2 = F.conv2d(x, w, None, 1, 1)
            rep = bool(torch.equal(y1, y2))
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                F.conv2d(x, w, None, 1, 1); torch.cuda.synchronize()
            names = [f"{e.key[:70]}:{e.device_time_total:.0f}us" for e in prof.key_averages() if e.device_time_total > 5]
        out.append(f"C={C} {t:.3f} ms {gf / t:.1f} TF repro={rep} [{' | '.join(names)}]")
    except Exception as e:
        out.append(f"C={C} FAILED {repr(e)[:120]}")
print(f"[{tag} {layout} det={int(det)}] " + "  ;  ".join(out), flush=True)
