"""SparseModule / SparseSequential (spconv/modules.py upstream; used at spconv_backbone.py:20,29,76-116)."""
from collections import OrderedDict

import torch
from torch import nn

from .tensor import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: modules that consume and produce a SparseConvTensor."""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    """Sequential container mixing sparse modules with dense ones (BatchNorm1d, ReLU, ...) that act on `.features`."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        if not torch.is_grad_enabled():
            return self._forward_inference(input)
        for module in self._modules.values():
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:        # BatchNorm1d cannot take an empty batch
                    input.features = module(input.features)
            else:
                input = module(input)
        return input


def _forward_inference(self, input):
    """no-grad path: a sparse convolution followed by an eval-mode BatchNorm1d (and a ReLU) runs as ONE launch with the
    normalisation folded into the weights and the activation in the GEMM epilogue (conv.forward_fused)."""
    from .conv import SparseConvolution
    mods = list(self._modules.values())
    i = 0
    while i < len(mods):
        module = mods[i]
        if isinstance(module, SparseConvolution) and isinstance(input, SparseConvTensor) and input.indices.shape[0] != 0:
            bn = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(bn, nn.BatchNorm1d) and not bn.training and bn.track_running_stats:
                relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                input = module.forward_fused(input, bn, relu)
                i += 3 if relu else 2
                continue
        if is_spconv_module(module):
            assert isinstance(input, SparseConvTensor)
            input = module(input)
        elif isinstance(input, SparseConvTensor):
            if input.indices.shape[0] != 0:
                input.features = module(input.features)
        else:
            input = module(input)
        i += 1
    return input


SparseSequential._forward_inference = _forward_inference


def prebuild_rulebooks(module, indices, spatial_shape, batch_size, indice_dict):
    """Walks `module` (SparseSequential nests, sparse convolutions, and composite SparseModules exposing their
    sparse children as attributes in execution order) and builds every rulebook from the coordinates alone.
    Returns the (indices, spatial_shape) leaving the module.  Dense layers (BatchNorm1d, ReLU) are skipped."""
    from .conv import SparseConvolution
    if isinstance(module, SparseConvolution):
        if indices.dtype != torch.int32:
            indices = indices.int()
        return module.build_rulebook(indices.contiguous(), spatial_shape, batch_size, indice_dict)
    if isinstance(module, SparseModule) or isinstance(module, nn.Sequential):
        for child in module._modules.values():
            if isinstance(child, (SparseModule, nn.Sequential)):
                indices, spatial_shape = prebuild_rulebooks(child, indices, spatial_shape, batch_size, indice_dict)
    return indices, spatial_shape
