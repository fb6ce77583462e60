"""csrc/wino_conv.hip — the stride-1 3x3 convolutions of BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:34-45:
Conv2d(3x3, padding 1, bias=False) + BatchNorm2d + ReLU) as Winograd F(2x2, 3x3) on the fp32 matrix cores.
Oracle: the direct convolution evaluated in float64 (torch CPU), which is what the reference's cuDNN fp32 convolution
approximates; tolerance 1e-4 of the output scale (north_star), written at each assert; measured error ~1e-6.
The reference-golden test of the whole folded backbone (tests/test_gpu_pointpillar_path.py::test_bev_backbone_and_box_decode_...)
runs through the same kernel once FoldedBEVBackbone routes its eligible layers here."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lidardetection_amd import _lib, wino

pytestmark = pytest.mark.gpu


def _ref(x, w, bias, relu):
    y = F.conv2d(x.double().cpu(), w.double().cpu(), None if bias is None else bias.double().cpu(), 1, 1)
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("B,cin,cout,H,W", [
    (2, 64, 64, 20, 36),       # 2 x 2 waves, partial blocks in both directions
    (1, 64, 64, 16, 16),       # exactly one workgroup
    (3, 128, 128, 14, 22),     # 1 x 4 waves
    (2, 256, 256, 9, 11),      # two channel groups per spatial block, odd sizes (bounds in the last tile row / column)
    (1, 16, 32, 7, 5),         # smallest supported channel counts: 4 x 1 waves, four chunks
    (2, 64, 128, 6, 40),       # Cin != Cout
    (1, 128, 64, 33, 17),
    (1, 64, 128, 62, 54),      # tall wave tiles (8 x 4 tiles) are chosen: 28 workgroups instead of 32 (PointPillar block 3 geometry)
    (1, 64, 64, 32, 8),        # tall, 2 x 2 waves
    (2, 16, 32, 64, 8),        # tall, 4 x 1 waves
])
def test_wino_conv3x3_vs_float64_direct_convolution(dev, B, cin, cout, H, W):
    g = torch.Generator(device="cpu").manual_seed(1000 * cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g)
    x[:, :, 0, :] += 2.0                                    # make the borders matter (zero padding must really be zero)
    x[:, :, :, -1] -= 3.0
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    bias = torch.randn(cout, generator=g)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    packed = wino.pack_weights(w.to(dev))
    for relu, b in ((True, bias), (False, None), (False, bias)):
        want = _ref(x, w, b, relu)
        got = wino.conv3x3(xd, packed, cout, None if b is None else b.to(dev), relu)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        scale = max(1.0, float(want.abs().max()))
        err = float((got.double().cpu() - want).abs().max())
        assert err <= 1e-4 * scale, (err, scale)            # north_star tolerance; typically 2e-6
    # channels-last weights (what FoldedBEVBackbone holds) pack to the same filters
    assert torch.equal(wino.pack_weights(w.to(dev).contiguous(memory_format=torch.channels_last)), packed)


def test_wino_conv3x3_writes_its_channel_slice_only(dev):
    """out_off / out_C: the layer's channels inside a wider NHWC map (base_bev_backbone.py:103 concat), neighbours untouched"""
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(2, 64, 12, 20, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    bias = torch.randn(64, generator=g)
    out = torch.full((2, 160, 12, 20), 7.0, device=dev).contiguous(memory_format=torch.channels_last)
    wino.conv3x3(x.to(dev).contiguous(memory_format=torch.channels_last), wino.pack_weights(w.to(dev)), 64, bias.to(dev), True, out=out, out_offset=32)
    want = _ref(x, w, bias, True)
    assert float((out[:, 32:96].double().cpu() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    assert bool((out[:, :32] == 7.0).all()) and bool((out[:, 96:] == 7.0).all())


def test_wino_conv3x3_is_deterministic_and_matches_miopen(dev):
    """run-to-run bit-identical (fixed summation order, no atomics — the library's split-K kernels are not), and within 1e-4 of
    the stock fp32 convolution on a backbone-sized map (PointPillar block 2: 128 channels, 124 x 108)"""
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.randn(4, 128, 124, 108, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, generator=g) / np.sqrt(9 * 128)).to(dev)
    bias = torch.randn(128, generator=g).to(dev)
    packed = wino.pack_weights(w)
    a = wino.conv3x3(x, packed, 128, bias, True)
    b = wino.conv3x3(x, packed, 128, bias, True)
    assert torch.equal(a, b)
    ref = torch.relu(F.conv2d(x, w.contiguous(memory_format=torch.channels_last), bias, 1, 1))
    assert float((a - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_wino_boundary_rejects_unsupported_shapes(dev):
    assert not wino.supported(4, 32) and not wino.supported(64, 48) and wino.supported(16, 32)
    assert not wino.supported(8, 32)                     # fewer than four chunks per block: the cross-block input pipeline needs them
    assert wino.supported43(32, 64) and not wino.supported43(16, 64) and not wino.supported43(64, 32) and not wino.supported43(40, 64)
    with pytest.raises(_lib.LidarHipError):
        wino.pack_weights(torch.zeros(48, 64, 3, 3, device=dev))
    x = torch.zeros(1, 64, 8, 8, device=dev).contiguous(memory_format=torch.channels_last)
    with pytest.raises(_lib.LidarHipError):
        wino.conv3x3(x, torch.zeros(16, device=dev), 64)
    with pytest.raises(_lib.LidarHipError):
        wino.conv3x3(torch.zeros(1, 64, 8, 8, device=dev), wino.pack_weights(torch.zeros(64, 64, 3, 3, device=dev)), 64)   # NCHW strides


def test_wino_two_waves_per_simd_kernel_matches(dev):
    """LIDAR_WINO_X2=1 (csrc/wino_conv.hip wino_f23x2_kernel: the 16 positions split between a wave pair on one SIMD, partial output
    transforms met in LDS) — measured no faster than the default kernel and therefore off, but kept under test: run in a child
    process (the switch is read once per process) and compare against the float64 direct convolution at the same 1e-4."""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np, torch, torch.nn.functional as F
from lidardetection_amd import wino
dev = torch.device('cuda:0')
for B, cin, cout, H, W in ((2, 64, 64, 20, 36), (1, 128, 128, 14, 22), (1, 64, 256, 62, 54), (1, 64, 64, 32, 8)):
    g = torch.Generator(device='cpu').manual_seed(cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g); w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin); b = torch.randn(cout, generator=g)
    got = wino.conv3x3(x.to(dev).contiguous(memory_format=torch.channels_last), wino.pack_weights(w.to(dev)), cout, b.to(dev), True)
    want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1))
    err = float((got.double().cpu() - want).abs().max())
    assert err <= 1e-4 * max(1.0, float(want.abs().max())), (cin, cout, H, W, err)
print('x2 ok')
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LIDAR_WINO_X2="1", PYTHONPATH=root), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "x2 ok" in r.stdout, r.stderr[-800:]


def test_wino_grouped_conv_vs_float64(dev):
    """lidar_wino_conv3x3_grouped_nhwc: n independent 3x3 convolutions in one launch, group g reading its own 64-channel slice of the
    input and writing output channels [32 g, 32 g + 32) (AnchorHeadMulti's second-layer branch convolutions, pcdet/models/dense_heads/
    anchor_head_multi.py:60-110: 2..12 real output channels per branch, filters zero-padded to 32) against per-branch float64 convolutions."""
    g = torch.Generator(device="cpu").manual_seed(21)
    B, H, W, cin, couts = 2, 18, 26, 64, [4, 12, 2, 6, 32, 10, 8]
    n = len(couts)
    x = torch.randn(B, n * cin + 8, H, W, generator=g)                          # a wider map: 8 trailing channels nobody reads
    ws = [torch.randn(c, cin, 3, 3, generator=g) / 24.0 for c in couts]
    bs = [torch.randn(c, generator=g) for c in couts]
    w_all, b_all = torch.zeros(32 * n, cin, 3, 3), torch.zeros(32 * n)
    for k, (w, b) in enumerate(zip(ws, bs)):
        w_all[32 * k:32 * k + w.shape[0]], b_all[32 * k:32 * k + w.shape[0]] = w, b
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    out = wino.conv3x3_grouped(xd, wino.pack_weights(w_all.to(dev)), cin, n, b_all.to(dev), False)
    assert out.shape == (B, 32 * n, H, W) and out.is_contiguous(memory_format=torch.channels_last)
    for k, (w, b) in enumerate(zip(ws, bs)):
        want = F.conv2d(x[:, k * cin:(k + 1) * cin].double(), w.double(), b.double(), 1, 1)
        got = out[:, 32 * k:32 * k + w.shape[0]].double().cpu()
        assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), k
        assert not out[:, 32 * k + w.shape[0]:32 * (k + 1)].any()                 # padded output channels: zero filters, zero bias
    with pytest.raises(_lib.LidarHipError):
        wino.conv3x3_grouped(xd, wino.pack_weights(w_all.to(dev)), cin, n + 1)   # more groups than the input holds


def test_wino_grouped_compact_output(dev):
    """the compact form writes only each group's real channels, back to back: equal to slicing the padded result"""
    g = torch.Generator(device="cpu").manual_seed(22)
    B, H, W, cin, couts = 1, 10, 14, 64, [2, 4, 2, 6, 4, 4, 8, 8, 4, 12, 8, 8, 5]
    n = len(couts)
    x = torch.randn(B, n * cin, H, W, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w_all, b_all = torch.zeros(32 * n, cin, 3, 3), torch.zeros(32 * n)
    for k, c in enumerate(couts):
        w_all[32 * k:32 * k + c] = torch.randn(c, cin, 3, 3, generator=g) / 24.0
        b_all[32 * k:32 * k + c] = torch.randn(c, generator=g)
    packed = wino.pack_weights(w_all.to(dev))
    padded = wino.conv3x3_grouped(x, packed, cin, n, b_all.to(dev), True)
    out, tables = wino.conv3x3_grouped_compact(x, packed, cin, couts, b_all.to(dev), True)
    out2, _ = wino.conv3x3_grouped_compact(x, packed, cin, couts, b_all.to(dev), True, tables)
    assert out.shape == (B, sum(couts), H, W) and torch.equal(out, out2)
    off = 0
    for k, c in enumerate(couts):
        assert torch.equal(out[:, off:off + c], padded[:, 32 * k:32 * k + c]), k
        off += c


# ------------------------------------------------------------------ F(4x4, 3x3): csrc/wino43_conv.hip
@pytest.mark.parametrize("B,cin,cout,H,W", [
    (1, 32, 64, 16, 16),       # one tile group pair, four chunks of 8 channels (the fewest the kernel takes)
    (2, 64, 64, 20, 36),       # partial groups in both directions
    (1, 64, 64, 32, 16),       # exactly one workgroup of 4 x 4-tile groups
    (3, 128, 128, 14, 22),     # two channel groups
    (2, 256, 256, 9, 11),      # four channel groups, odd sizes (bounds inside the last tile row / column)
    (2, 64, 128, 6, 40),       # 2 x 8-tile groups are chosen
    (1, 128, 64, 33, 17),
    (1, 64, 128, 62, 54),      # 8 x 2-tile groups (PointPillar block 3 geometry)
    (2, 32, 64, 70, 7),        # four chunks per block, several blocks per workgroup: the input stream crosses blocks mid-pipeline
    (1, 64, 64, 124, 108),     # several blocks per workgroup on one XCD share (persistent walk)
])
def test_wino_f43_conv3x3_vs_float64_direct_convolution(dev, B, cin, cout, H, W):
    """F(4x4, 3x3), points {0, 1, -1, 1/2, -2, inf}: larger transform constants than F(2x2) -> |error| ~ 1e-5 of the output scale;
    the assert is the north_star tolerance 1e-4, the measured maximum is printed by -s"""
    g = torch.Generator(device="cpu").manual_seed(1000 * cin + cout + H)
    x = torch.randn(B, cin, H, W, generator=g)
    x[:, :, 0, :] += 2.0
    x[:, :, :, -1] -= 3.0
    w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    bias = torch.randn(cout, generator=g)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    packed = wino.pack_weights43(w.to(dev))
    for relu, b in ((True, bias), (False, None), (False, bias)):
        want = _ref(x, w, b, relu)
        got = wino.conv3x3_f43(xd, packed, cout, None if b is None else b.to(dev), relu)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        scale = max(1.0, float(want.abs().max()))
        err = float((got.double().cpu() - want).abs().max())
        print(f"f43 {B}x{cin}->{cout} {H}x{W}: max err {err:.2e} (scale {scale:.1f})")
        assert err <= 1e-4 * scale, (err, scale)
    assert torch.equal(wino.pack_weights43(w.to(dev).contiguous(memory_format=torch.channels_last)), packed)


def test_wino_f43_slice_in_and_out_and_determinism(dev):
    """reads channels [0, Cin) of a wider input map, writes its slice of a wider output map (neighbours untouched), bit-identical
    run to run"""
    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.randn(2, 96, 23, 41, generator=g)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    bias = torch.randn(64, generator=g)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    packed = wino.pack_weights43(w.to(dev))
    out = torch.full((2, 160, 23, 41), 7.0, device=dev).contiguous(memory_format=torch.channels_last)
    wino.conv3x3_f43(xd, packed, 64, bias.to(dev), True, out=out, out_offset=32, cin=64)
    want = _ref(x[:, :64], w, bias, True)
    assert float((out[:, 32:96].double().cpu() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    assert bool((out[:, :32] == 7.0).all()) and bool((out[:, 96:] == 7.0).all())
    out2 = torch.full_like(out, 7.0)
    wino.conv3x3_f43(xd, packed, 64, bias.to(dev), True, out=out2, out_offset=32, cin=64)
    assert torch.equal(out, out2)


def test_wino_f43_matches_f23_on_a_backbone_sized_map(dev):
    """PointPillar block 2 geometry (128 channels, 124 x 108, 4 frames): both Winograd kernels within 1e-4 of the library's direct
    fp32 convolution, and of each other"""
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.relu(torch.randn(4, 128, 124, 108, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, generator=g) / np.sqrt(9 * 128)).to(dev)
    bias = torch.randn(128, generator=g).to(dev)
    a = wino.conv3x3_f43(x, wino.pack_weights43(w), 128, bias, True)
    b = wino.conv3x3(x, wino.pack_weights(w), 128, bias, True)
    ref = torch.relu(F.conv2d(x, w.contiguous(memory_format=torch.channels_last), bias, 1, 1))
    scale = max(1.0, float(ref.abs().max()))
    print(f"f43 vs direct {float((a - ref).abs().max()):.2e}, f23 vs direct {float((b - ref).abs().max()):.2e}, scale {scale:.1f}")
    assert float((a - ref).abs().max()) <= 1e-4 * scale
    assert float((a - b).abs().max()) <= 1e-4 * scale


def test_wino_f43_random_small_shapes(dev):
    """24 seeded random shapes (maps from 1 x 1 up, channel counts from the smallest supported) against the fp64 convolution, through a
    wider input map and into a slice of a wider output map: the zero padding by buffer bounds, the bounds-dropped stores of partial
    tiles, the cross-block input stream with few chunks per block"""
    r = np.random.default_rng(4343)
    for case in range(24):
        B, H, W = int(r.integers(1, 4)), int(r.integers(1, 41)), int(r.integers(1, 41))
        cin, cout = int(r.choice([32, 48, 64, 96])), int(r.choice([64, 128]))
        in_c, out_c, off = cin + 4 * int(r.integers(0, 3)), cout + 32, 4 * int(r.integers(0, 9))
        g = torch.Generator(device="cpu").manual_seed(100 + case)
        x = torch.randn(B, in_c, H, W, generator=g)
        w = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
        bias = torch.randn(cout, generator=g)
        out = torch.full((B, out_c, H, W), -5.0, device=dev).contiguous(memory_format=torch.channels_last)
        wino.conv3x3_f43(x.to(dev).contiguous(memory_format=torch.channels_last), wino.pack_weights43(w.to(dev)), cout, bias.to(dev), bool(case & 1),
                         out=out, out_offset=off, cin=cin)
        want = _ref(x[:, :cin], w, bias, bool(case & 1))
        got = out[:, off:off + cout].double().cpu()
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) <= 1e-4 * scale, (case, B, H, W, cin, cout)
        assert bool((out[:, :off] == -5.0).all()) and bool((out[:, off + cout:] == -5.0).all()), case
