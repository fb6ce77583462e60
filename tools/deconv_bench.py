"""The fused deblock GEMM (csrc/deconv_gemm.hip) on PointPillar's two strided deblocks, bs 16: ms and fraction of the fp32 MFMA peak.
LIDAR_HIP_SO selects an A/B build."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd.bev_backbone import deconv_pack, deconv_gemm_into_
dev = torch.device("cuda:0")


def ev(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for B, K, h, w, s, cup in [(16, 128, 124, 108, 2, 128), (16, 256, 62, 54, 4, 128), (16, 256, 100, 88, 2, 256)]:
    g = torch.Generator(device="cpu").manual_seed(K)
    x = torch.randn(B, K, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wk = (torch.randn(K, s * s * cup, generator=g) / K ** 0.5).to(dev)
    bias = torch.randn(cup, generator=g).to(dev)
    out = torch.empty((B, 384, s * h, s * w), device=dev).contiguous(memory_format=torch.channels_last)
    pk = deconv_pack(wk)
    t = ev(lambda: deconv_gemm_into_(x, pk, bias, s, out, 128))
    gf = 2.0 * B * h * w * K * s * s * cup / 1e9
    print(f"K{K} s{s} {h}x{w} -> {cup}: {t:.3f} ms = {gf / t:.1f} TF = {gf / t / 157.3:.3f} of peak", flush=True)
