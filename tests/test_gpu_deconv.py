"""csrc/deconv_gemm.hip — the deblocks of BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:51-57: ConvTranspose2d with
kernel == stride, BatchNorm2d, ReLU; :103 torch.cat of the upsampled maps) as one fp32-MFMA kernel that writes straight into the
layer's channel slice of the concatenated NHWC map.  Oracle: torch's conv_transpose2d evaluated in float64 on the CPU (the
reference's op), tolerance 1e-4 of the output scale (north_star); neighbours of the slice must stay untouched."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lidardetection_amd import _lib
from lidardetection_amd import bev_backbone as bb

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,K,h,w,s,c_up,c_out,off", [
    (2, 128, 14, 22, 2, 128, 384, 128),     # PointPillar deblock 2 geometry (one 512-column group), 616 pixels: partial last block
    (1, 256, 9, 11, 4, 128, 384, 256),      # PointPillar deblock 3: four column groups
    (1, 256, 10, 12, 2, 256, 512, 256),     # SECOND / multi-head deblock 2: 256 up-channels (two 128-channel rounds per (ky, kx))
    (3, 64, 5, 7, 2, 128, 128, 0),          # K = 64 (16 chunks), fewer pixels than one block
    (1, 16, 16, 16, 2, 128, 132, 4),         # smallest K, 256 pixels = two blocks exactly, odd slice offset
])
def test_deconv_gemm_vs_float64_conv_transpose(dev, B, K, h, w, s, c_up, c_out, off):
    g = torch.Generator(device="cpu").manual_seed(K + 7 * s + c_up + h)
    x = torch.randn(B, K, h, w, generator=g)
    wt = torch.randn(K, c_up, s, s, generator=g) / np.sqrt(K)                     # ConvTranspose2d weight layout (Cin, Cout, kH, kW)
    bias = torch.randn(c_up, generator=g)
    assert bb.deconv_supported(K, s, c_up)
    w_kn = wt.permute(0, 2, 3, 1).reshape(K, -1).contiguous().to(dev)            # columns (ky, kx, c), as FoldedBEVBackbone folds it
    packed = bb.deconv_pack(w_kn)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    for relu in (True, False):
        out = torch.full((B, c_out, s * h, s * w), 7.0, device=dev).contiguous(memory_format=torch.channels_last)
        bb.deconv_gemm_into_(xd, packed, bias.to(dev), s, out, off, relu)
        want = F.conv_transpose2d(x.double(), wt.double(), bias.double(), stride=s)
        want = torch.relu(want) if relu else want
        err = float((out[:, off:off + c_up].double().cpu() - want).abs().max())
        assert err <= 1e-4 * max(1.0, float(want.abs().max())), (err, relu)
        assert bool((out[:, :off] == 7.0).all()) and bool((out[:, off + c_up:] == 7.0).all())


def test_deconv_gemm_equals_library_gemm_plus_shuffle_and_is_deterministic(dev):
    """against the r03 two-step path (library GEMM into a temporary + lidar_bias_act_upsample_nhwc) on the PointPillar deblock-3
    shape, and run-to-run bit-identical (fixed summation order)"""
    g = torch.Generator(device="cpu").manual_seed(3)
    B, K, h, w, s, c_up = 2, 256, 62, 54, 4, 128
    x = torch.randn(B, K, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w_kn = (torch.randn(K, s * s * c_up, generator=g) / 16.0).to(dev)
    bias = torch.randn(c_up, generator=g).to(dev)
    packed = bb.deconv_pack(w_kn)
    a = torch.zeros((B, 384, s * h, s * w), device=dev).contiguous(memory_format=torch.channels_last)
    b = torch.zeros_like(a)
    ref = torch.zeros_like(a)
    bb.deconv_gemm_into_(x, packed, bias, s, a, 256)
    bb.deconv_gemm_into_(x, packed, bias, s, b, 256)
    assert torch.equal(a, b)
    y = bb.rows_gemm(x.permute(0, 2, 3, 1).reshape(B * h * w, K), w_kn)
    bb.bias_act_upsample_(y, bias, B, h, w, s, ref, 256)
    assert float((a - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_deconv_boundary_rejects_unsupported_shapes(dev):
    assert not bb.deconv_supported(64, 1, 128) and not bb.deconv_supported(128, 2, 32) and not bb.deconv_supported(60, 2, 128)
    with pytest.raises(_lib.LidarHipError):
        bb.deconv_pack(torch.zeros(64, 128, device=dev))
    x = torch.zeros(1, 64, 4, 4, device=dev).contiguous(memory_format=torch.channels_last)
    out = torch.zeros(1, 128, 8, 8, device=dev)                                  # not channels-last
    with pytest.raises(_lib.LidarHipError):
        bb.deconv_gemm_into_(x, bb.deconv_pack(torch.zeros(64, 512, device=dev)), torch.zeros(128, device=dev), 2, out)
