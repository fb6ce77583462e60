import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def ev(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
x = cl(torch.randn(4, 64, 128, 128, device=dev))
w1 = cl(torch.randn(2304, 64, 3, 3, device=dev) * 0.05)
print("conv 64->2304      :", ev(lambda: F.conv2d(x, w1, None, padding=1)))
for parts in (2, 4, 6, 12, 36):
    ws = [cl(w) for w in w1.chunk(parts, 0)]
    print(f"conv 64->{2304 // parts} x {parts:2d}  :", ev(lambda: [F.conv2d(x, w, None, padding=1) for w in ws]))
y = cl(torch.randn(4, 2304, 128, 128, device=dev))
w2 = cl(torch.randn(432, 64, 3, 3, device=dev) * 0.05)
print("grouped 36 x (64->12):", ev(lambda: F.conv2d(y, w2, None, padding=1, groups=36)))
y6 = [cl(torch.randn(4, 384, 128, 128, device=dev)) for _ in range(6)]
w26 = cl(torch.randn(72, 64, 3, 3, device=dev) * 0.05)
print("grouped 6 x (64->12) x 6 heads:", ev(lambda: [F.conv2d(t, w26, None, padding=1, groups=6) for t in y6]))
ys = [cl(torch.randn(4, 64, 128, 128, device=dev)) for _ in range(36)]
w2s = [cl(torch.randn(8, 64, 3, 3, device=dev) * 0.05) for _ in range(36)]
print("36 separate 64->8  :", ev(lambda: [F.conv2d(a, b, None, padding=1) for a, b in zip(ys, w2s)]))
