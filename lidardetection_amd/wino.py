"""Host side of csrc/wino_conv.hip: the stride-1 3x3 convolutions of the dense BEV backbone (pcdet/models/backbones_2d/
base_bev_backbone.py:34-45) as Winograd F(2x2, 3x3) on the fp32 matrix cores, shift + ReLU in the kernel's epilogue."""
import os

import torch

from . import _lib

# which Winograd kernel the model code gets for a layer both support: 1 (default) = F(4x4, 3x3) (csrc/wino43_conv.hip: fewer MFMA
# cycles, |error| ~ 1e-5 of the output scale), 0 = F(2x2, 3x3) everywhere (csrc/wino_conv.hip: ~ 2e-6).  A/B switch.
_F43 = [os.environ.get("LIDAR_WINO_F43", "1") != "0"]


def supported(cin, cout):
    return bool(_lib.lib().lidar_wino_supported(int(cin), int(cout)))


def pack_weights(w):
    """w (Cout, Cin, 3, 3) fp32 (BatchNorm scale folded in) -> the packed transformed filters the kernel reads (16 Cin Cout floats)"""
    _lib.require_cuda(w.contiguous())
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or w.dtype != torch.float32 or not supported(w.shape[1], w.shape[0]):
        raise _lib.LidarHipError(f"wino.pack_weights: expected a float32 (Cout % 32 == 0, Cin % 8 == 0, Cin >= 16, 3, 3) weight, got {tuple(w.shape)}")
    wc = w.detach().contiguous()                      # plain (Cout, Cin, 3, 3) order whatever the memory format of `w`
    if wc.stride() != (wc.shape[1] * 9, 9, 3, 1):
        wc = wc.clone(memory_format=torch.contiguous_format)
    L = _lib.lib()
    packed = torch.empty(L.lidar_wino_packed_floats(w.shape[1], w.shape[0]), dtype=torch.float32, device=w.device)
    _lib.check(L.lidar_wino_pack_weights(_lib.ptr(wc), w.shape[1], w.shape[0], _lib.ptr(packed), _lib.stream()), "lidar_wino_pack_weights")
    return packed


def conv3x3(x, packed, cout, bias=None, relu=True, out=None, out_offset=0):
    """x (B, Cin, H, W) channels-last fp32 -> act(conv3x3(x, w, padding=1) + bias) as a channels-last (B, cout, H, W) tensor, or
    into channels [out_offset, out_offset + cout) of the channels-last `out` (B, C_out, H, W)."""
    _lib.require_cuda(packed, bias)
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.LidarHipError("wino.conv3x3: expected a channels-last float32 CUDA tensor")
    B, cin, H, W = x.shape
    L = _lib.lib()
    if packed.numel() != L.lidar_wino_packed_floats(cin, cout) or packed.numel() == 0:
        raise _lib.LidarHipError("wino.conv3x3: packed filters do not match (Cin, Cout)")
    if bias is not None and bias.numel() != cout:
        raise _lib.LidarHipError("wino.conv3x3: bias must hold Cout values")
    if out is None:
        out, out_offset = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last), 0
    elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous(memory_format=torch.channels_last)
              and out.shape[0] == B and tuple(out.shape[2:]) == (H, W) and 0 <= out_offset and out_offset + cout <= out.shape[1]):
        raise _lib.LidarHipError("wino.conv3x3: output must be channels-last (B, C_out, H, W) with room for the slice")
    _lib.check(L.lidar_wino_conv3x3_nhwc(_lib.ptr(x), B, H, W, cin, _lib.ptr(packed), _lib.ptr(bias), int(bool(relu)), int(cout),
                                         _lib.ptr(out), out.shape[1], int(out_offset), _lib.stream()), "lidar_wino_conv3x3_nhwc")
    return out


def supported43(cin, cout):
    """the F(4x4, 3x3) kernel (csrc/wino43_conv.hip) takes this layer: Cin % 16 == 0, Cin >= 32, Cout % 64 == 0"""
    return bool(_lib.lib().lidar_wino43_supported(int(cin), int(cout)))


def pack_weights43(w):
    """w (Cout, Cin, 3, 3) fp32 -> the packed F(4x4, 3x3) filters (36 Cin Cout floats; the filter transform runs in fp64)"""
    _lib.require_cuda(w.contiguous())
    if w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or w.dtype != torch.float32 or not supported43(w.shape[1], w.shape[0]):
        raise _lib.LidarHipError(f"wino.pack_weights43: expected a float32 (Cout % 64 == 0, Cin % 16 == 0, 3, 3) weight, got {tuple(w.shape)}")
    wc = w.detach().contiguous()
    if wc.stride() != (wc.shape[1] * 9, 9, 3, 1):
        wc = wc.clone(memory_format=torch.contiguous_format)
    L = _lib.lib()
    packed = torch.empty(L.lidar_wino43_packed_floats(w.shape[1], w.shape[0]), dtype=torch.float32, device=w.device)
    _lib.check(L.lidar_wino43_pack_weights(_lib.ptr(wc), w.shape[1], w.shape[0], _lib.ptr(packed), _lib.stream()), "lidar_wino43_pack_weights")
    return packed


def conv3x3_f43(x, packed, cout, bias=None, relu=True, out=None, out_offset=0, cin=None):
    """conv3x3 through the F(4x4, 3x3) kernel.  cin: the layer reads channels [0, cin) of x (default: all of them)."""
    _lib.require_cuda(packed, bias)
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.LidarHipError("wino.conv3x3_f43: expected a channels-last float32 CUDA tensor")
    B, in_c, H, W = x.shape
    cin = in_c if cin is None else int(cin)
    L = _lib.lib()
    if cin > in_c or packed.numel() != L.lidar_wino43_packed_floats(cin, cout) or packed.numel() == 0:
        raise _lib.LidarHipError("wino.conv3x3_f43: packed filters do not match (Cin, Cout)")
    if bias is not None and bias.numel() != cout:
        raise _lib.LidarHipError("wino.conv3x3_f43: bias must hold Cout values")
    if out is None:
        out, out_offset = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last), 0
    elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous(memory_format=torch.channels_last)
              and out.shape[0] == B and tuple(out.shape[2:]) == (H, W) and 0 <= out_offset and out_offset + cout <= out.shape[1]):
        raise _lib.LidarHipError("wino.conv3x3_f43: output must be channels-last (B, C_out, H, W) with room for the slice")
    _lib.check(L.lidar_wino43_conv3x3_nhwc(_lib.ptr(x), B, H, W, cin, in_c, _lib.ptr(packed), _lib.ptr(bias), int(bool(relu)), int(cout),
                                           _lib.ptr(out), out.shape[1], int(out_offset), _lib.stream()), "lidar_wino43_conv3x3_nhwc")
    return out


def pack_auto(w):
    """-> [kind, packed filters, weight]: F(4x4, 3x3) where csrc/wino43_conv.hip takes the layer (and LIDAR_WINO_F43 != 0), else
    F(2x2, 3x3).  The weight rides along: a map too large for the F(4x4) kernel's 32-bit byte offsets (>= 2 GiB) is served by F(2x2),
    packed on first need."""
    if _F43[0] and supported43(w.shape[1], w.shape[0]):
        return ["f43", pack_weights43(w), w.detach()]
    return ["f23", pack_weights(w), None]


def conv3x3_auto(x, packed, cout, bias=None, relu=True, out=None, out_offset=0):
    """conv3x3 with the filters of pack_auto"""
    kind, p = packed[0], packed[1]
    if kind == "f43":
        if x.numel() * 4 < 2 ** 31 - 1 and (out is None or out.numel() * 4 < 2 ** 31 - 1):
            return conv3x3_f43(x, p, cout, bias, relu, out, out_offset)
        if len(packed) < 4:                            # oversize map: F(2x2) filters, packed once
            packed.append(pack_weights(packed[2]))
        p = packed[3]
    return conv3x3(x, p, cout, bias, relu, out, out_offset)


def conv3x3_grouped_compact(x, packed, group_cin, couts, bias=None, relu=False, tables=None):
    """conv3x3_grouped with only the REAL output channels written: -> (B, sum(couts), H, W) channels-last, group g at channels
    [sum(couts[:g]), + couts[g]).  tables = (grp_cout, grp_ooff) device int32 tensors from a previous call (returned as second value)."""
    _lib.require_cuda(packed, bias)
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.LidarHipError("wino.conv3x3_grouped_compact: expected a channels-last float32 CUDA tensor")
    B, C, H, W = x.shape
    n = len(couts)
    L = _lib.lib()
    if n * group_cin > C or packed.numel() != L.lidar_wino_packed_floats(group_cin, 32 * n) or packed.numel() == 0 or max(couts) > 32 or min(couts) < 1:
        raise _lib.LidarHipError("wino.conv3x3_grouped_compact: groups / packed filters do not match the input")
    if bias is not None and bias.numel() != 32 * n:
        raise _lib.LidarHipError("wino.conv3x3_grouped_compact: bias must hold 32 * n_groups values (padded like the filters)")
    if tables is None:
        offs = [0]
        for c in couts[:-1]:
            offs.append(offs[-1] + int(c))
        tables = (torch.tensor([int(c) for c in couts], dtype=torch.int32, device=x.device), torch.tensor(offs, dtype=torch.int32, device=x.device))
    ctot = int(sum(couts))
    out = torch.empty((B, ctot, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _lib.check(L.lidar_wino_conv3x3_grouped_compact_nhwc(_lib.ptr(x), B, H, W, C, int(group_cin), n, _lib.ptr(packed), _lib.ptr(bias),
                                                         int(bool(relu)), _lib.ptr(tables[0]), _lib.ptr(tables[1]), _lib.ptr(out), ctot, 0,
                                                         _lib.stream()), "lidar_wino_conv3x3_grouped_compact_nhwc")
    return out, tables


def conv3x3_grouped(x, packed, group_cin, n_groups, bias=None, relu=False, out=None, out_offset=0):
    """n_groups independent 3x3 / padding-1 convolutions in one launch: group g reads channels [g * group_cin, (g + 1) * group_cin) of the
    channels-last x (B, C >= n_groups * group_cin, H, W) and writes channels [32 g, 32 g + 32) of the result (B, 32 n_groups, H, W).
    packed = pack_weights of the stacked (32 n_groups, group_cin, 3, 3) filters (groups with fewer outputs: zero rows)."""
    _lib.require_cuda(packed, bias)
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.LidarHipError("wino.conv3x3_grouped: expected a channels-last float32 CUDA tensor")
    B, C, H, W = x.shape
    cout = 32 * int(n_groups)
    L = _lib.lib()
    if n_groups * group_cin > C or packed.numel() != L.lidar_wino_packed_floats(group_cin, cout) or packed.numel() == 0:
        raise _lib.LidarHipError("wino.conv3x3_grouped: groups / packed filters do not match the input")
    if bias is not None and bias.numel() != cout:
        raise _lib.LidarHipError("wino.conv3x3_grouped: bias must hold 32 * n_groups values")
    if out is None:
        out, out_offset = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last), 0
    elif not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous(memory_format=torch.channels_last)
              and out.shape[0] == B and tuple(out.shape[2:]) == (H, W) and 0 <= out_offset and out_offset + cout <= out.shape[1]):
        raise _lib.LidarHipError("wino.conv3x3_grouped: output must be channels-last (B, C_out, H, W) with room for the slice")
    _lib.check(L.lidar_wino_conv3x3_grouped_nhwc(_lib.ptr(x), B, H, W, C, int(group_cin), int(n_groups), _lib.ptr(packed), _lib.ptr(bias),
                                                 int(bool(relu)), _lib.ptr(out), out.shape[1], int(out_offset), _lib.stream()),
               "lidar_wino_conv3x3_grouped_nhwc")
    return out
