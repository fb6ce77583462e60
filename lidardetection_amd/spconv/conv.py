"""SubMConv3d / SparseConv3d / SparseInverseConv3d (spconv/conv.py upstream; call sites SURVEY.md §2d)."""
import math

import torch
from torch import nn
from torch.nn import init

from . import ops
from .modules import SparseModule
from .tensor import SparseConvTensor


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

    def build_rulebook(self, indices, spatial_shape, batch_size, indice_dict):
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
                nbr = ops.subm_rulebook(indices, spatial_shape, self.kernel_size)
                datas = {"subm": True, "nbr": nbr, "nbr_t": nbr, "in_indices": indices, "out_indices": indices,
                         "in_spatial_shape": spatial_shape, "out_spatial_shape": spatial_shape}
                if self.indice_key is not None:
                    indice_dict[self.indice_key] = datas
            return indices, spatial_shape
        out_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding)
        if datas is None:
            out_indices, nbr, nbr_t = ops.conv_rulebook(indices, batch_size, spatial_shape, self.kernel_size, self.stride, self.padding)
            datas = {"subm": False, "nbr": nbr, "nbr_t": nbr_t, "in_indices": indices, "out_indices": out_indices,
                     "in_spatial_shape": spatial_shape, "out_spatial_shape": out_shape}
            if self.indice_key is not None:
                indice_dict[self.indice_key] = datas
        return datas["out_indices"], out_shape

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        features, indices = input.features, input.indices
        spatial_shape, batch_size = input.spatial_shape, input.batch_size
        if indices.dtype != torch.int32:
            indices = indices.int()
        indices = indices.contiguous()
        if self.subm:
            out_shape = spatial_shape
        elif self.inverse:
            out_shape = None
        else:
            out_shape = ops.get_conv_output_size(spatial_shape, self.kernel_size, self.stride, self.padding)

        if self.conv1x1 and not self.inverse:
            f = torch.mm(features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                f = f + self.bias
            out = SparseConvTensor(f, indices, spatial_shape, batch_size)
            out.indice_dict, out.grid = input.indice_dict, input.grid
            return out

        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            assert datas is not None and self.indice_key is not None, "inverse conv needs the rulebook of its paired conv"
            out_indices, out_shape = datas["in_indices"], datas["in_spatial_shape"]
            fwd_table, bwd_table, flip = datas["nbr_t"], datas["nbr"], False
            assert datas["out_indices"].shape[0] == indices.shape[0], "inverse conv input does not match the paired conv's output"
        elif self.subm:
            if datas is None:
                nbr = ops.subm_rulebook(indices, spatial_shape, self.kernel_size)
                datas = {"subm": True, "nbr": nbr, "nbr_t": nbr, "in_indices": indices, "out_indices": indices,
                         "in_spatial_shape": spatial_shape, "out_spatial_shape": spatial_shape}
                if self.indice_key is not None:
                    input.indice_dict[self.indice_key] = datas
            out_indices = indices
            fwd_table, bwd_table, flip = datas["nbr"], datas["nbr"], True
        else:
            if datas is None:
                out_indices, nbr, nbr_t = ops.conv_rulebook(indices, batch_size, spatial_shape, self.kernel_size, self.stride,
                                                            self.padding)
                datas = {"subm": False, "nbr": nbr, "nbr_t": nbr_t, "in_indices": indices, "out_indices": out_indices,
                         "in_spatial_shape": spatial_shape, "out_spatial_shape": out_shape}
                if self.indice_key is not None:
                    input.indice_dict[self.indice_key] = datas
            out_indices = datas["out_indices"]
            fwd_table, bwd_table, flip = datas["nbr"], datas["nbr_t"], False

        out_features = ops.indice_conv(features, self.weight, self.bias, fwd_table, bwd_table, flip)
        out = SparseConvTensor(out_features, out_indices, out_shape, batch_size)
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
