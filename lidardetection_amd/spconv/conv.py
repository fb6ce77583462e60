"""SubMConv3d / SparseConv3d / SparseInverseConv3d (spconv/conv.py upstream; call sites SURVEY.md §2d)."""
import math

import torch
from torch import nn
from torch.nn import init

from . import ops
from .modules import SparseModule
from .tensor import SparseConvTensor


_SIDE_STREAMS = {}


def _schedule_mask_order(datas, key, table):
    """Inference: compute the mask order of a freshly built table on a side stream (tiny, latency-bound kernels) so it
    overlaps the remaining rulebook builds; consumers wait on the recorded event (forward_fused).  The outputs are allocated
    on the current stream and the kernels handed the side stream's raw handle (no stream switch on the host)."""
    dev = table.device
    main = torch.cuda.current_stream(dev)
    side = _SIDE_STREAMS.get(dev)
    if side is None:
        side = _SIDE_STREAMS[dev] = torch.cuda.Stream(dev, priority=-1)        # short, latency-critical kernels: ahead of the GEMMs
    side.wait_stream(main)                      # the table is produced on the main stream
    import ctypes
    masks, perm = ops.mask_order(table, ctypes.c_void_p(side.cuda_stream))
    ev = torch.cuda.Event()
    ev.record(side)
    for t in (table, masks, perm):
        t.record_stream(side)
    datas[key] = [masks, perm, ev]


def _triple(v):
    return [int(v)] * 3 if isinstance(v, int) else [int(x) for x in v]


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None):
        super().__init__()
        assert ndim == 3 and groups == 1
        self.ndim, self.in_channels, self.out_channels = ndim, in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _triple(kernel_size), _triple(stride), _triple(padding)
        self.dilation = _triple(dilation)
        assert self.dilation == [1, 1, 1], "dilated sparse convolutions are not used by the reference"
        assert not transposed, "SparseConvTranspose3d is not used by the reference"
        self.conv1x1 = all(k == 1 for k in self.kernel_size)
        self.subm, self.inverse, self.indice_key = subm, inverse, indice_key
        self.weight = nn.Parameter(torch.Tensor(*self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def needs_new_rulebook(self, indice_dict):
        if self.conv1x1 or self.inverse:
            return False
        return self.indice_key is None or self.indice_key not in indice_dict

    def begin_rulebook(self, indices, spatial_shape, batch_size, indice_dict=None):
        """regular (strided) conv only: start the output-site search early (ops.conv_rulebook_begin)"""
        assert not self.subm and not self.inverse and not self.conv1x1
        return ops.conv_rulebook_begin(indices, batch_size, spatial_shape, self.kernel_size, self.stride, self.padding, indice_dict)

    def build_rulebook(self, indices, spatial_shape, batch_size, indice_dict, pending=None):
        """Coordinate-only part of forward(): makes sure this layer's rulebook is in `indice_dict` and returns the
        (indices, spatial_shape) of its output.  Rulebooks depend on coordinates alone, so a network can build all of
        them up front (the only host read-backs of the sparse path) and then run its feature pass without a single sync."""
        if self.conv1x1 and not self.inverse:
            return indices, spatial_shape
        datas = indice_dict.get(self.indice_key) if self.indice_key is not None else None
        if self.inverse:
            assert datas is not None, "inverse conv needs the rulebook of its paired conv"
            return datas["in_indices"], datas["in_spatial_shape"]
        if self.subm:
            if datas is None:
                nbr = ops.subm_rulebook(indices, spatial_shape, self.kernel_size, batch_size, indice_dict)
                datas = {"subm": True, "nbr": nbr, "nbr_t": nbr, "in_indices": indices, "out_indices": indices,
                         "in_spatial_shape": spatial_shape, "out_spatial_shape": spatial_shape}
                if self.indice_key is not None:
                    indice_dict[self.indice_key] = datas
                self._maybe_schedule_order(datas)
            return indices, spatial_shape
        out_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding)
        if datas is None:
            if pending is None:
                pending = self.begin_rulebook(indices, spatial_shape, batch_size, indice_dict)
            out_indices, nbr, nbr_t = ops.conv_rulebook_finish(pending, indice_dict.get("__spec__"),
                                                               need_t=not indice_dict.get("__inference__", False))
            datas = {"subm": False, "nbr": nbr, "nbr_t": nbr_t, "in_indices": indices, "out_indices": out_indices,
                     "in_spatial_shape": spatial_shape, "out_spatial_shape": out_shape}
            if self.indice_key is not None:
                indice_dict[self.indice_key] = datas
            self._maybe_schedule_order(datas)
        return datas["out_indices"], out_shape

    def _maybe_schedule_order(self, datas):
        """called right after this layer built a new table inside a prebuild pass"""
        nbr = datas["nbr"]
        if nbr.is_cuda and nbr.shape[0] > 0 and nbr.shape[1] <= 31 and (
                self.indice_key is not None                 # shared table: a later layer of the level will want the order
                or ops.sorted_gemm_supported(nbr.shape[1], self.in_channels, self.out_channels)):
            _schedule_mask_order(datas, "order", nbr)

    def _resolve(self, input, need_bwd=True):
        """-> (indices, out_indices, out_shape, fwd_table, bwd_table, flip, datas) for this layer on `input` (rulebook built
        or fetched through indice_key); tables are None for a 1x1 convolution; datas = the rulebook dict used (returned,
        not kept on the module: a module may be entered again before an earlier call has finished).  need_bwd=False: the
        caller will not read bwd_table (inference), so a transposed table the builder skipped is not made up now."""
        indices = input.indices
        spatial_shape, batch_size = input.spatial_shape, input.batch_size
        if indices.dtype != torch.int32:
            indices = indices.int()
        indices = indices.contiguous()
        if self.conv1x1 and not self.inverse:
            return indices, indices, spatial_shape, None, None, False, None
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert datas is not None and self.indice_key is not None, "inverse conv needs the rulebook of its paired conv"
            assert datas["out_indices"].shape[0] == indices.shape[0], "inverse conv input does not match the paired conv's output"
            return indices, datas["in_indices"], datas["in_spatial_shape"], ops.ensure_table_t(datas), datas["nbr"], False, datas
        if self.subm:
            if datas is None:
                nbr = ops.subm_rulebook(indices, spatial_shape, self.kernel_size, batch_size, input.indice_dict)
                datas = {"subm": True, "nbr": nbr, "nbr_t": nbr, "in_indices": indices, "out_indices": indices,
                         "in_spatial_shape": spatial_shape, "out_spatial_shape": spatial_shape}
                if self.indice_key is not None:
                    input.indice_dict[self.indice_key] = datas
            return indices, indices, spatial_shape, datas["nbr"], datas["nbr"], True, datas
        out_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding)
        if datas is None:
            out_indices, nbr, nbr_t = ops.conv_rulebook(indices, batch_size, spatial_shape, self.kernel_size, self.stride,
                                                        self.padding, input.indice_dict)
            datas = {"subm": False, "nbr": nbr, "nbr_t": nbr_t, "in_indices": indices, "out_indices": out_indices,
                     "in_spatial_shape": spatial_shape, "out_spatial_shape": out_shape}
            if self.indice_key is not None:
                input.indice_dict[self.indice_key] = datas
        return indices, datas["out_indices"], out_shape, datas["nbr"], (ops.ensure_table_t(datas) if need_bwd else None), False, datas

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        indices, out_indices, out_shape, fwd_table, bwd_table, flip, datas = self._resolve(input)
        if fwd_table is None:
            f = torch.mm(input.features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                f = f + self.bias
        else:
            # mask orders are cached per table in the rulebook dict: "order" belongs to datas["nbr"], "order_t" to datas["nbr_t"]
            keys = ("order_t", "order") if self.inverse else (("order", "order") if self.subm else ("order", "order_t"))
            orders = (datas, *keys) if (input.features.is_cuda and fwd_table.shape[1] <= 31) else None
            f = ops.indice_conv(input.features, self.weight, self.bias, fwd_table, bwd_table, flip, orders)
        out = SparseConvTensor(f, out_indices, out_shape, input.batch_size)
        out.indice_dict, out.grid = input.indice_dict, input.grid
        return out

    # ---- inference fast path: conv + BatchNorm1d(eval) (+ residual) (+ ReLU) in one launch -------------------------
    def _folded(self, bn):
        """(weight (K, Cin, Cout) * bn scale, bn shift (+ scaled conv bias)); cached until a parameter changes."""
        srcs = [self.weight, self.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
        key = tuple((t.data_ptr(), t._version) for t in srcs if t is not None) + (id(bn),)
        cache = getattr(self, "_fold_cache", None)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w = self.weight.detach().reshape(-1, self.in_channels, self.out_channels)
                b = self.bias.detach() if self.bias is not None else None
                if bn is not None:
                    scale = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
                    shift = bn.bias.detach() - bn.running_mean * scale
                    w = w * scale.view(1, 1, -1)
                    b = shift if b is None else b * scale + shift
                w = w.contiguous()
                packed = ops.pack_gemm_weights(w) if (w.is_cuda and ops.sorted_gemm_supported(w.shape[0], self.in_channels, self.out_channels)) else None
                cache = (key, w, None if b is None else b.contiguous(), packed)
            self._fold_cache = cache
        return cache[1], cache[2]

    def _folded_packed(self):
        """the packed form of the last _folded() weights (None: no packed kernel for this shape)"""
        cache = getattr(self, "_fold_cache", None)
        return None if cache is None else cache[3]

    def forward_fused(self, input, bn=None, relu=False, residual=None):
        """act(bn(conv(input)) + residual) with eval-mode BatchNorm1d folded into the weights; no autograd graph.
        Equals the unfused module sequence to fp32 rounding (one re-associated multiply)."""
        assert isinstance(input, SparseConvTensor)
        assert bn is None or (not bn.training and bn.track_running_stats), "BatchNorm must be in eval mode to be folded"
        indices, out_indices, out_shape, fwd_table, _, _, datas = self._resolve(input, need_bwd=False)
        w, b = self._folded(bn)
        feats = input.features.detach().contiguous()
        if fwd_table is None:
            f = torch.mm(feats, w[0]) if b is None else torch.addmm(b, feats, w[0])
            if residual is not None:
                f = f + residual
            if relu:
                f = torch.relu_(f)
        else:
            st = None
            if ops.sorted_gemm_supported(w.shape[0], self.in_channels, self.out_channels):
                # mask order of the table's rows, shared through indice_key
                st = ops.cached_mask_order(datas, "order_t" if self.inverse else "order", fwd_table)
            f = ops.indice_conv_fused(feats, fwd_table, w, b, None if residual is None else residual.contiguous(), relu, st,
                                      self._folded_packed() if st is not None else None)
        out = SparseConvTensor(f, out_indices, out_shape, input.batch_size)
        out.indice_dict, out.grid = input.indice_dict, input.grid
        return out


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        # stride / padding arguments are accepted and ignored: submanifold convs keep the input sites (Appendix A.2)
        super().__init__(3, in_channels, out_channels, kernel_size, 1, 0, dilation, groups, bias, True, indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super().__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True, indice_key=indice_key)
