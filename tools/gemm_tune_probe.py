"""The four plain GEMMs of the PointPillar BEV backbone (three deblocks + merged heads): hipBLASLt's default pick through torch
vs. PyTorch TunableOp's pick (tuned in-process)."""
import os, sys, time
import torch
dev = torch.device("cuda:0")
shapes = {"deblock1 (fused elsewhere)": (16 * 248 * 216, 64, 128), "deblock2": (16 * 124 * 108, 128, 512), "deblock3": (16 * 62 * 54, 256, 2048),
          "heads (addmm)": (16 * 248 * 216, 384, 72)}


def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ops = {}
for name, (M, K, N) in shapes.items():
    A = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev); b = torch.randn(N, device=dev)
    ops[name] = (lambda A=A, W=W, b=b: torch.addmm(b, A, W)) if "heads" in name else (lambda A=A, W=W: torch.mm(A, W))
base = {k: t(f) for k, f in ops.items()}
torch.cuda.tunable.enable(True)
torch.cuda.tunable.tuning_enable(True)
try:
    torch.cuda.tunable.set_max_tuning_duration(200)
    torch.cuda.tunable.set_max_tuning_iterations(20)
except Exception as e:
    print("tunable limits:", e)
t0 = time.time()
tuned = {k: t(f) for k, f in ops.items()}
print(f"tuning took {time.time() - t0:.1f} s")
for k in ops:
    M, K, N = shapes[k]
    print(f"{k:28s} M {M} K {K} N {N}: default {base[k]:7.1f} us ({2 * M * K * N / base[k] * 1e-6:6.1f} TF)   tuned {tuned[k]:7.1f} us ({2 * M * K * N / tuned[k] * 1e-6:6.1f} TF)")
