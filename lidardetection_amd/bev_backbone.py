"""Inference fast path of the dense BEV backbone + anchor head.

The convolutions stay stock torch (MIOpen, fp32, channels-last).  What changes is everything around them:
  * eval-mode BatchNorm2d is folded into the preceding convolution (scale -> weights, shift -> bias) — the
    reference's Conv2d/BN/ReLU triplets (pcdet/models/backbones_2d/base_bev_backbone.py:34-45) and
    ConvTranspose2d/BN/ReLU deblocks (base_bev_backbone.py:51-57);
  * shift + ReLU run as ONE in-place HIP pass (`lidar_bias_act_nhwc`) instead of a BN pass and a ReLU pass;
  * the deblock epilogues write directly into their channel slice of the concatenated map
    (base_bev_backbone.py:103), so `torch.cat` disappears;
  * ConvTranspose2d with kernel == stride (every deblock of the reference configs) is evaluated as one plain GEMM on
    the NHWC map (rocBLAS/hipBLASLt through torch.mm) and the epilogue does the pixel shuffle;
  * the three 1x1 heads of AnchorHeadSingle (pcdet/models/dense_heads/anchor_head_single.py:18-33,45-55) read the
    384-channel map once (one merged GEMM, torch.addmm) instead of three times.
Results match the unfolded modules to fp32 rounding (folding re-associates one multiply); tests assert 1e-4.
"""
import ctypes as C
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, wino, workspace


def bias_act_(x, bias, relu=True, out=None, out_offset=0):
    """x: (B, C, H, W) channels-last fp32.  out=None: in place.  Otherwise out is a channels-last (B, C_out, H, W)
    tensor and the result lands in channels [out_offset, out_offset + C)."""
    _lib.require_cuda(bias)
    if not (x.is_cuda and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.LidarHipError("bias_act_: expected a channels-last CUDA tensor")
    B, C, H, W = x.shape
    if out is None:
        out, out_offset = x, 0
    elif not (out.is_contiguous(memory_format=torch.channels_last) and out.shape[0] == B and out.shape[2:] == x.shape[2:]):
        raise _lib.LidarHipError("bias_act_: output must be channels-last with the same batch / spatial shape")
    _lib.check(_lib.lib().lidar_bias_act_nhwc(_lib.ptr(x), _lib.ptr(bias), B * H * W, C, int(bool(relu)), _lib.ptr(out),
                                              out.shape[1], int(out_offset), _lib.stream()), "lidar_bias_act_nhwc")
    return out


_LT_GEMM = [os.environ.get("LIDAR_BEV_LT_GEMM", "1") != "0"]      # the fused stride-1 deblock (csrc/dense_gemm.hip); A/B switch
_SPLIT_MIN = [int(os.environ.get("LIDAR_BEV_SPLIT_MIN", "8"))]    # fewest frames per part-batch
_SPLIT = [int(os.environ.get("LIDAR_BEV_SPLIT", "2"))]            # part-batches / streams of FoldedBEVBackbone.merged (1 = off)
_WINO = [os.environ.get("LIDAR_BEV_WINO", "1") != "0"]            # stride-1 3x3 layers on csrc/wino_conv.hip (0: MIOpen + epilogue pass)
_DECONV = [os.environ.get("LIDAR_BEV_DECONV", "1") != "0"]        # stride > 1 deblocks on csrc/deconv_gemm.hip (0: library GEMM + pixel-shuffle pass)
_SPARSE_FIRST = [os.environ.get("LIDAR_BEV_SPARSE_FIRST", "1") != "0"]   # first layer straight from the pillars (0: dense canvas + MIOpen)


def _lt_gemm(a_ptr, M, K, w_kn, bias, relu, d_ptr, ldd, device):
    """D (M rows at pitch ldd) = act(A (M, K) @ w_kn (K, N) + bias): lidar_dense_gemm_bias_act (hipBLASLt, candidates timed once
    per shape).  False: the library path is not available (nothing written)."""
    ws = workspace.get("dense_gemm", 32 << 20, device)
    st = _lib.lib().lidar_dense_gemm_bias_act(a_ptr, M, K, _lib.ptr(w_kn), w_kn.shape[1], _lib.ptr(bias), int(bool(relu)), d_ptr, ldd,
                                              _lib.ptr(ws), ws.numel(), _lib.stream())
    if st == -4:                                   # LIDAR_ERR_UNSUPPORTED
        return False
    _lib.check(st, "lidar_dense_gemm_bias_act")
    return True


def _side_streams(device, n):
    """the n (<= 2) side streams of the split forward: the SAME two streams the sparse forward uses for its mask orders and rulebooks
    (spconv/conv.py, spconv/modules.py — never busy at the same time as a dense backbone).  A process drives only a few hardware
    queues (4 by default); streams beyond them share queues and serialise falsely: with each subsystem owning its own streams
    (7 in bench.py's `extra` process) the three-stream sparse forward measured there went from 3.15 to 4.4 ms."""
    from .spconv import conv as _sc, modules as _sm
    dev = torch.device(device)
    out = []
    for pool in (_sc._SIDE_STREAMS, _sm._RULEBOOK_STREAMS)[:n]:
        st = pool.get(dev)
        if st is None:
            st = pool[dev] = torch.cuda.Stream(dev, priority=-1)
        out.append(st)
    return out


def gemm_bias_act_into_(x, w_kn, bias, out, out_offset, relu=True):
    """x (B, K, h, w) channels-last, w_kn (K, N), bias (N): out[:, out_offset:out_offset+N] = act(x_rows @ w_kn + bias) with `out`
    (B, C_out, h, w) channels-last — one hipBLASLt GEMM whose epilogue writes at the map's row pitch.
    Returns False when the library path is not available (nothing written)."""
    _lib.require_cuda(w_kn, bias)
    if not (x.is_cuda and out.is_cuda and x.dtype == torch.float32 and out.dtype == torch.float32):
        raise _lib.LidarHipError("gemm_bias_act_into_: expected float32 CUDA (ROCm) tensors")
    B, K, h, w = x.shape
    N = w_kn.shape[1]
    if (not x.is_contiguous(memory_format=torch.channels_last) or not out.is_contiguous(memory_format=torch.channels_last)
            or out.shape[0] != B or out.shape[2:] != x.shape[2:] or w_kn.shape[0] != K or bias.numel() != N
            or out_offset + N > out.shape[1] or not w_kn.is_contiguous()):
        raise _lib.LidarHipError("gemm_bias_act_into_: shapes / layouts do not match")
    return _lt_gemm(_lib.ptr(x), B * h * w, K, w_kn, bias, relu, C.c_void_p(out.data_ptr() + 4 * out_offset), out.shape[1], x.device)


def rows_gemm(a2d, w_kn, bias=None, out=None):
    """a2d (M, K) @ w_kn (K, N) (+ bias) -> (M, N) (into `out`, contiguous, when given): the library GEMM with its candidates timed
    once per shape; torch.mm / addmm when that path is not available."""
    if _LT_GEMM[0] and a2d.is_cuda and a2d.dtype == torch.float32 and a2d.is_contiguous() and w_kn.is_contiguous():
        if out is None:
            out = torch.empty((a2d.shape[0], w_kn.shape[1]), dtype=torch.float32, device=a2d.device)
        elif not out.is_contiguous() or out.shape != (a2d.shape[0], w_kn.shape[1]):
            raise _lib.LidarHipError("rows_gemm: out must be a contiguous (M, N) tensor")
        if _lt_gemm(_lib.ptr(a2d), a2d.shape[0], a2d.shape[1], w_kn, bias, False, _lib.ptr(out), w_kn.shape[1], a2d.device):
            return out
        _LT_GEMM[0] = False
    if out is None:
        return torch.mm(a2d, w_kn) if bias is None else torch.addmm(bias, a2d, w_kn)
    return torch.mm(a2d, w_kn, out=out) if bias is None else torch.addmm(bias, a2d, w_kn, out=out)


def deconv_supported(K, s, c_up):
    return bool(_lib.lib().lidar_deconv_supported(int(K), int(s), int(c_up)))


def deconv_pack(w_kn):
    """(K, s*s*C_up) folded ConvTranspose2d weight, columns (ky, kx, c) -> packed form of csrc/deconv_gemm.hip"""
    _lib.require_cuda(w_kn)
    K, N = w_kn.shape
    L = _lib.lib()
    n = L.lidar_deconv_packed_floats(K, N)
    if n == 0:
        raise _lib.LidarHipError(f"deconv_pack: unsupported shape ({K}, {N})")
    packed = torch.empty(n, dtype=torch.float32, device=w_kn.device)
    _lib.check(L.lidar_deconv_pack_weights(_lib.ptr(w_kn.contiguous()), K, N, _lib.ptr(packed), _lib.stream()), "lidar_deconv_pack_weights")
    return packed


def deconv_gemm_into_(x, packed, bias, s, out, out_offset=0, relu=True):
    """x (B, K, h, w) channels-last -> out[:, out_offset:out_offset + C_up] (B, C_out, s*h, s*w channels-last) = act(ConvTranspose2d
    (kernel == stride == s)(x) + bias): one fp32-MFMA kernel, no temporary, no pixel-shuffle pass (csrc/deconv_gemm.hip)"""
    _lib.require_cuda(packed, bias)
    B, K, h, w = x.shape
    c_up = bias.numel()
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous(memory_format=torch.channels_last) and out.is_cuda
            and out.dtype == torch.float32 and out.is_contiguous(memory_format=torch.channels_last) and out.shape[0] == B
            and tuple(out.shape[2:]) == (s * h, s * w) and 0 <= out_offset and out_offset + c_up <= out.shape[1]
            and packed.numel() == K * s * s * c_up):
        raise _lib.LidarHipError("deconv_gemm_into_: shapes / layouts do not match")
    _lib.check(_lib.lib().lidar_deconv_gemm_nhwc(_lib.ptr(x), B, h, w, K, _lib.ptr(packed), _lib.ptr(bias), int(bool(relu)), int(s), c_up,
                                                 _lib.ptr(out), out.shape[1], int(out_offset), _lib.stream()), "lidar_deconv_gemm_nhwc")
    return out


def bias_act_upsample_(y2d, bias, batch, h, w, s, out, out_offset=0, relu=True):
    """y2d: (batch*h*w, s*s*C) GEMM output of a kernel==stride transposed conv, columns ordered (ky, kx, c).
    Writes act(y + bias) pixel-shuffled into channels [out_offset, out_offset + C) of the channels-last `out`."""
    _lib.require_cuda(y2d, bias)
    C = bias.numel()
    if y2d.shape != (batch * h * w, s * s * C):
        raise _lib.LidarHipError("bias_act_upsample_: GEMM output shape does not match (batch*h*w, s*s*C)")
    if not (out.is_contiguous(memory_format=torch.channels_last) and out.shape[0] == batch
            and tuple(out.shape[2:]) == (h * s, w * s)):
        raise _lib.LidarHipError("bias_act_upsample_: output must be channels-last (batch, C_out, h*s, w*s)")
    _lib.check(_lib.lib().lidar_bias_act_upsample_nhwc(_lib.ptr(y2d), _lib.ptr(bias), batch, h, w, s, C, int(bool(relu)),
                                                       _lib.ptr(out), out.shape[1], int(out_offset), _lib.stream()),
               "lidar_bias_act_upsample_nhwc")
    return out


def collect_params(*modules):
    """every parameter and buffer of `modules` (collected once: the module tree does not change)"""
    out = []
    for m in modules:
        out += list(m.parameters()) + list(m.buffers())
    return out


def params_key(tensors):
    """identity + version of the source tensors: changes whenever load_state_dict(), .to(), an optimizer step or a
    BatchNorm statistics update touches one of them (same idea as SparseConvolution._folded)"""
    return tuple((t.data_ptr(), t._version) for t in tensors)


def _fold(weight, bn, out_dim, conv_bias=None):
    """scale the conv weight along its output-channel dim by gamma/sqrt(var+eps); return (weight, shift).
    A convolution bias (the reference uses bias=False) is folded as bias*scale + shift."""
    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias.detach() - bn.running_mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach() * scale
    shape = [1] * weight.dim()
    shape[out_dim] = -1
    return (weight.detach() * scale.view(shape)), shift.contiguous()


class FoldedBEVBackbone:
    """Built from the (eval-mode) reference-shaped modules; call with the channels-last canvas."""

    def __init__(self, blocks, deblocks, heads):
        self.sources = collect_params(blocks, deblocks, *heads)
        self.source_key = params_key(self.sources)                  # owners rebuild when this no longer matches
        self.stages = []
        for blk, de in zip(blocks, deblocks):
            convs, mods, i = [], list(blk), 0
            while i < len(mods):
                pad = 0
                if isinstance(mods[i], nn.ZeroPad2d):
                    pad, i = int(mods[i].padding[0]), i + 1
                conv, bn = mods[i], mods[i + 1]
                assert isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and isinstance(mods[i + 2], nn.ReLU)
                w, b = _fold(conv.weight, bn, 0, conv.bias)
                # stride-1 3x3 layers (LAYER_NUMS per block, base_bev_backbone.py:40-45; SECOND's first block opens with one too):
                # Winograd on the matrix cores with shift + ReLU in the kernel (csrc/wino43_conv.hip F(4x4, 3x3), csrc/wino_conv.hip F(2x2, 3x3))
                packed = None
                if (_WINO[0] and w.is_cuda and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1)
                        and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.padding[0] + pad == 1
                        and conv.padding[1] + pad == 1 and wino.supported(w.shape[1], w.shape[0])):
                    packed = wino.pack_auto(w)        # F(4x4, 3x3) where supported (Cout % 64 == 0), else F(2x2, 3x3)
                convs.append((w.contiguous(memory_format=torch.channels_last), b, conv.stride, conv.padding[0] + pad, packed))
                i += 3
            up, bn = de[0], de[1]
            if isinstance(up, nn.ConvTranspose2d):
                w, b = _fold(up.weight, bn, 1, up.bias)
                if tuple(up.kernel_size) == tuple(up.stride) and up.stride[0] == up.stride[1] and \
                        tuple(up.padding) == (0, 0) and tuple(up.output_padding) == (0, 0):
                    # kernel == stride: every input pixel owns its own s x s output patch -> a plain GEMM
                    w_kn = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
                    upc = ("gemm", w_kn, b, up.stride[0])
                    if (_DECONV[0] and w.is_cuda and up.stride[0] > 1 and deconv_supported(w_kn.shape[0], up.stride[0], b.numel())):
                        upc = ("deconv_mfma", (deconv_pack(w_kn), w_kn), b, up.stride[0])   # csrc/deconv_gemm.hip (+ the plain weight: maps of 2 GiB and more take the two-step path)
                else:
                    upc = ("deconv", w.contiguous(memory_format=torch.channels_last), b, up.stride)
            else:   # stride < 1 in the reference config: a strided Conv2d (base_bev_backbone.py:60-69)
                w, b = _fold(up.weight, bn, 0, up.bias)
                upc = ("conv", w.contiguous(memory_format=torch.channels_last), b, up.stride)
            self.stages.append((convs, upc))
        self.up_channels = [s[1][2].numel() for s in self.stages]
        for h in heads:
            assert tuple(h.kernel_size) == (1, 1) and tuple(h.stride) == (1, 1)
        self.head_split = [h.weight.shape[0] for h in heads]
        if heads:   # (a backbone used without merged 1x1 heads — e.g. AnchorHeadMulti's 3x3 branches — calls features() only)
            self.head_wt = torch.cat([h.weight.detach().flatten(1) for h in heads], 0).t().contiguous()   # (C_in, sum C_head)
            self.head_b = torch.cat([h.bias.detach() if h.bias is not None else h.weight.new_zeros(h.weight.shape[0])
                                     for h in heads], 0).contiguous()
        self._cats = {}                    # concat buffers, one per (slot of a split forward, shape)
        self._streams = None
        # the first layer as a sparse implicit GEMM over the pillars (csrc/pillar.hip lidar_pillar_conv_table): (k*k, Cin, Cout) weights
        self._first_sparse = None
        w0, b0, st0, pad0, _ = self.stages[0][0][0]
        k0 = w0.shape[2]
        if (w0.is_cuda and w0.shape[2] == w0.shape[3] and st0[0] == st0[1] and k0 * k0 <= 31):
            from .spconv import ops as _sops
            if _sops.sorted_gemm_supported(k0 * k0, w0.shape[1], w0.shape[0]):
                self._first_sparse = (w0.permute(2, 3, 1, 0).reshape(k0 * k0, w0.shape[1], w0.shape[0]).contiguous(), b0, k0, int(st0[0]), int(pad0))

    def first_layer_from_pillars(self, pm):
        """pm: pillar_ops.PillarMap -> act(conv0(scatter(pm)) + shift) as a channels-last (B, Cout, OH, OW) map WITHOUT building the
        canvas: neighbour table over all output pixels (one tiny launch) + the mask-ordered sparse implicit GEMM, which writes every
        output pixel once (pixels no pillar reaches get act(shift)).  PointPillar-KITTI bs 16: 4.7 GFLOP instead of 63."""
        from . import pillar_ops
        from .spconv import ops as _sops
        w9, b0, k0, st0, pad0 = self._first_sparse
        nbr, OH, OW = pillar_ops.pillar_conv_table(pm.coords, pm.B, pm.nx, pm.ny, k0, st0, pad0, pm.num_voxels_dev)
        order = _sops.mask_order(nbr)
        out = _sops.indice_conv_fused(pm.features, nbr, w9, b0, None, True, order)           # (B * OH * OW, Cout) = the NHWC map
        return out.view(pm.B, OH, OW, w9.shape[2]).permute(0, 3, 1, 2)

    def sparse_first_ok(self):
        return self._first_sparse is not None and _SPARSE_FIRST[0]

    def features(self, canvas, slot=0, first_done=False):
        """-> the concatenated upsampled map (B, sum(up_channels), H, W), channels-last.  first_done: `canvas` is already the output
        of the first layer (first_layer_from_pillars)."""
        x, cat, off = canvas, None, 0
        for si, (convs, (kind, uw, ub, ustride)) in enumerate(self.stages):
            for ci, (w, b, stride, pad, packed) in enumerate(convs):
                if first_done and si == 0 and ci == 0:
                    continue
                if packed is not None and _WINO[0] and x.is_contiguous(memory_format=torch.channels_last):
                    x = wino.conv3x3_auto(x, packed, w.shape[0], b, True)
                    continue
                x = F.conv2d(x, w, None, stride, pad)
                if not x.is_contiguous(memory_format=torch.channels_last):
                    x = x.contiguous(memory_format=torch.channels_last)
                bias_act_(x, b)
            B, _, h, w = x.shape
            y = None
            if kind == "deconv_mfma":
                oh, ow = h * ustride, w * ustride
                if B * oh * ow * sum(self.up_channels) * 4 >= 2 ** 31 - 1:       # the fused kernel stores with 32-bit byte offsets
                    kind, uw = "gemm", uw[1]
                    y = rows_gemm(x.permute(0, 2, 3, 1).reshape(B * h * w, -1), uw)
                else:
                    uw = uw[0]
            elif kind == "gemm":
                oh, ow = h * ustride, w * ustride
                if not (ustride == 1 and _LT_GEMM[0]):
                    y = rows_gemm(x.permute(0, 2, 3, 1).reshape(B * h * w, -1), uw)  # the NHWC map IS the row-major A
            else:
                y = F.conv_transpose2d(x, uw, None, ustride) if kind == "deconv" else F.conv2d(x, uw, None, ustride)
                if not y.is_contiguous(memory_format=torch.channels_last):
                    y = y.contiguous(memory_format=torch.channels_last)
                oh, ow = y.shape[2], y.shape[3]
            if cat is None:
                shape = (B, sum(self.up_channels), oh, ow)
                cat = self._cats.get(slot)
                if cat is None or cat.shape != shape or cat.device != x.device:
                    cat = self._cats[slot] = torch.empty(shape, dtype=torch.float32, device=x.device,
                                                         memory_format=torch.channels_last)
            if kind == "gemm" and y is None:
                # stride 1: GEMM + shift + ReLU + concat in ONE hipBLASLt call writing with the map's row pitch
                if not gemm_bias_act_into_(x, uw, ub, cat, off):
                    _LT_GEMM[0] = False                                             # not available here: the two-step path from now on
                    y = rows_gemm(x.permute(0, 2, 3, 1).reshape(B * h * w, -1), uw)
            if kind == "deconv_mfma":
                deconv_gemm_into_(x, uw, ub, ustride, cat, off)
            elif kind == "gemm":
                if y is not None:
                    bias_act_upsample_(y, ub, B, h, w, ustride, cat, off)
            else:
                bias_act_(y, ub, out=cat, out_offset=off)
            off += ub.numel()
        return cat

    def stale(self):
        return params_key(self.sources) != self.source_key

    def merged(self, canvas):
        """-> the merged head output (B, H, W, sum C_head), channels [cls | box | dir] as the heads were given.
        canvas: the channels-last BEV map, or a pillar_ops.PillarMap (the first layer then runs from the pillars when it can)."""
        first_done = False
        if not torch.is_tensor(canvas):
            if self.sparse_first_ok():
                canvas, first_done = self.first_layer_from_pillars(canvas), True
            else:
                canvas = canvas.dense()
        B = canvas.shape[0]
        if 1 < _SPLIT[0] <= 2 and canvas.is_cuda and B % _SPLIT[0] == 0 and B // _SPLIT[0] >= _SPLIT_MIN[0]:   # (4-frame parts lost in r03: PV-RCNN bs 8 16.0 -> 17.5 ms)
            return self._merged_split(canvas, _SPLIT[0], first_done)
        cat = self.features(canvas, first_done=first_done)
        B, C, H, W = cat.shape
        out = rows_gemm(cat.permute(0, 2, 3, 1).reshape(B * H * W, C), self.head_wt, self.head_b)     # 1x1 heads = one GEMM
        return out.view(B, H, W, -1)

    def _merged_split(self, canvas, n, first_done=False):
        """The same forward as n part-batches on n streams: the convolutions are bound by the matrix cores (0.8 of the fp32 MFMA peak,
        < 0.5 TB/s of HBM traffic) and everything around them — MIOpen's zero-fill of each output, the shift + ReLU pass, the pixel
        shuffles — by HBM, and within one batch they are strictly serial; two part-batches let one's passes run under the other's
        convolutions (PointPillar bs 16: backbone + heads 9.94 -> 9.48 ms, step 10.52 -> 10.18 ms; SECOND bs 16: 18.49 -> 17.81 ms;
        8 frames alone cost 5.08 ms).  Same layers, same weights;
        the library may pick other kernels for the smaller batch, so results can differ from the whole-batch forward in the last bits."""
        dev = canvas.device
        cur = torch.cuda.current_stream(dev)
        self._streams = _side_streams(dev, n)
        B, hb = canvas.shape[0], canvas.shape[0] // n
        out = None
        for i, st in enumerate(self._streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                cat = self.features(canvas[i * hb:(i + 1) * hb], slot=i + 1, first_done=first_done)
                _, C, H, W = cat.shape
                if out is None:
                    with torch.cuda.stream(cur):                       # owned by the caller's stream, written by the side streams
                        out = torch.empty((B, H, W, self.head_wt.shape[1]), dtype=torch.float32, device=dev)
                    st.wait_stream(cur)
                out.record_stream(st)
                rows_gemm(cat.permute(0, 2, 3, 1).reshape(hb * H * W, C), self.head_wt, self.head_b, out=out[i * hb:(i + 1) * hb].view(hb * H * W, -1))
        for st in self._streams:
            cur.wait_stream(st)
        return out

    def __call__(self, canvas):
        """-> per-head maps in (B, H, W, C_head) layout (views of the merged head output)."""
        return torch.split(self.merged(canvas), self.head_split, dim=-1)
