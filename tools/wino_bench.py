"""Winograd 3x3 convolution (csrc/wino_conv.hip) vs the stock MIOpen convolution + this repo's shift / ReLU pass on the stride-1
layer shapes of the BEV backbones (PointPillar bs 16: 64 @ 248x216, 128 @ 124x108, 256 @ 62x54; SECOND bs 16: 128 @ 200x176,
256 @ 100x88).  Prints per shape: ms, direct-convolution-equivalent TFLOP/s (2 * 9 * Cin * Cout * pixels), MFMA-issued TFLOP/s
(the 16 / 36 of it that Winograd really multiplies) as a fraction of the 157.3 TFLOP/s fp32 peak, max error vs MIOpen."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import wino
from lidardetection_amd.bev_backbone import bias_act_
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True


def ev(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


shapes = [(16, 64, 248, 216), (16, 128, 124, 108), (16, 256, 62, 54), (16, 128, 200, 176), (16, 256, 100, 88)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for B, C, H, W in shapes:
    g = torch.Generator(device="cpu").manual_seed(C)
    x = torch.randn(B, C, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(dev)
    wl = w.contiguous(memory_format=torch.channels_last)
    bias = torch.randn(C, generator=g).to(dev)
    packed = wino.pack_weights(w)
    with torch.no_grad():
        t_w = ev(lambda: wino.conv3x3(x, packed, C, bias, True))
        if os.environ.get("WINO_BENCH_ONLY"):
            if wino.supported43(C, C):
                p43 = wino.pack_weights43(w)
                t_4 = ev(lambda: wino.conv3x3_f43(x, p43, C, bias, True))
                e43 = float((wino.conv3x3_f43(x, p43, C, bias, True) - wino.conv3x3(x, packed, C, bias, True)).abs().max())
                print(f"B{B} C{C} {H}x{W}: F(4x4) {t_4:.3f} ms, MFMA {2.0 * 9 * C * C * B * H * W / 1e9 / 4 / t_4 / 157.3:.3f} of peak, "
                      f"{t_w / t_4:.2f}x over F(2x2), max |diff| to it {e43:.2e} |", flush=True)
            print(f"B{B} C{C} {H}x{W}: wino {t_w:.3f} ms, MFMA {2.0 * 9 * C * C * B * H * W / 1e9 * 16 / 36 / t_w / 157.3:.3f} of peak |", flush=True)
            continue
        t_m = ev(lambda: bias_act_(F.conv2d(x, wl, None, 1, 1), bias))
        t_c = ev(lambda: F.conv2d(x, wl, None, 1, 1))
        err = float((wino.conv3x3(x, packed, C, bias, True) - bias_act_(F.conv2d(x, wl, None, 1, 1), bias)).abs().max())
    gf = 2.0 * 9 * C * C * B * H * W / 1e9
    print(f"B{B} C{C} {H}x{W}: wino {t_w:.3f} ms = {gf / t_w:.1f} TF direct-equivalent, MFMA {gf * 16 / 36 / t_w:.1f} TF = {gf * 16 / 36 / t_w / 157.3:.3f} of peak"
          f" | MIOpen conv {t_c:.3f} + epilogue = {t_m:.3f} ms ({gf / t_c:.1f} TF) | speed-up {t_m / t_w:.2f}x | max |diff| {err:.2e}", flush=True)
