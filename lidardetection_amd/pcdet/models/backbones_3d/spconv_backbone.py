"""The reference's two sparse 3D backbones — VoxelBackBone8x and VoxelResBackBone8x (pcdet/models/backbones_3d/
spconv_backbone.py:68-261), with its helper names `post_act_block` (:7-26) and `SparseBasicBlock` (:29-65) — built on
lidardetection_amd.spconv.  The stage layout is written as data (`_PLAIN_STAGES` / `_RES_STAGES`) and assembled by one
builder; module and parameter names come out exactly as the reference's, so its checkpoints (weight layout
(kD, kH, kW, Cin, Cout)) load unchanged (`tests/test_cabi_and_host.py` compares the state_dict layouts).

forward(): all rulebooks of the network are built first from the coordinates alone (the strided convs' output-count
read-backs overlap the SubM table builds), then the feature pass runs without a host sync; under torch.no_grad() every
conv + BatchNorm (+ residual) + ReLU is one launch (spconv.conv.forward_fused).
"""
from functools import partial

import torch
import torch.nn as nn

from .... import spconv

_CONV_KINDS = {
    'subm': lambda cin, cout, k, stride, padding, key: spconv.SubMConv3d(cin, cout, k, bias=False, indice_key=key),
    'spconv': lambda cin, cout, k, stride, padding, key: spconv.SparseConv3d(cin, cout, k, stride=stride, padding=padding,
                                                                             bias=False, indice_key=key),
    'inverseconv': lambda cin, cout, k, stride, padding, key: spconv.SparseInverseConv3d(cin, cout, k, indice_key=key, bias=False),
}


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm', norm_fn=None):
    """sparse conv of the requested kind + norm + ReLU as one SparseSequential"""
    if conv_type not in _CONV_KINDS:
        raise NotImplementedError(conv_type)
    conv = _CONV_KINDS[conv_type](in_channels, out_channels, kernel_size, stride, padding, indice_key)
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    """two 3x3x3 submanifold convs with an identity (or `downsample`) shortcut"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        if norm_fn is None:
            raise AssertionError('SparseBasicBlock needs a norm_fn')
        make_conv = partial(spconv.SubMConv3d, kernel_size=3, stride=stride, padding=1, bias=True, indice_key=indice_key)
        self.conv1, self.bn1 = make_conv(inplanes, planes), norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2, self.bn2 = make_conv(planes, planes), norm_fn(planes)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        if not torch.is_grad_enabled() and not self.bn1.training and x.indices.shape[0] != 0:
            # inference: each conv + BN (+ shortcut) + ReLU is one launch
            mid = self.conv1.forward_fused(x, self.bn1, relu=True)
            return self.conv2.forward_fused(mid, self.bn2, relu=True, residual=shortcut.features)
        out = self.conv1(x)
        out.features = self.relu(self.bn1(out.features))
        out = self.conv2(out)
        out.features = self.relu(self.bn2(out.features) + shortcut.features)
        return out


# stage name -> list of layers; ('down', cin, cout, padding) = strided SparseConv3d block, ('subm', c) = SubM block,
# ('res', c) = SparseBasicBlock.  Strided blocks of stage N use indice_key spconvN, the others submN / resN.
_PLAIN_STAGES = {
    'conv1': [('subm', 16)],
    'conv2': [('down', 16, 32, 1), ('subm', 32), ('subm', 32)],
    'conv3': [('down', 32, 64, 1), ('subm', 64), ('subm', 64)],
    'conv4': [('down', 64, 64, (0, 1, 1)), ('subm', 64), ('subm', 64)],
}
_RES_STAGES = {
    'conv1': [('res', 16), ('res', 16)],
    'conv2': [('down', 16, 32, 1), ('res', 32), ('res', 32)],
    'conv3': [('down', 32, 64, 1), ('res', 64), ('res', 64)],
    'conv4': [('down', 64, 128, (0, 1, 1)), ('res', 128), ('res', 128)],
}
_STAGE_ORDER = ('conv_input', 'conv1', 'conv2', 'conv3', 'conv4', 'conv_out')


class _VoxelBackBoneBase(nn.Module):
    def _assemble(self, model_cfg, input_channels, grid_size, stages, out_in_channels):
        self.model_cfg = model_cfg
        norm = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        zyx = [int(v) for v in list(grid_size)[::-1]]
        self.sparse_shape = [zyx[0] + 1, zyx[1], zyx[2]]          # one extra z slice, as the reference allocates
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm(16), nn.ReLU())
        for name, layers in stages.items():
            level = name[-1]
            built = []
            for spec in layers:
                if spec[0] == 'down':
                    _, cin, cout, pad = spec
                    built.append(post_act_block(cin, cout, 3, norm_fn=norm, stride=2, padding=pad, indice_key='spconv' + level,
                                                conv_type='spconv'))
                elif spec[0] == 'subm':
                    built.append(post_act_block(spec[1], spec[1], 3, norm_fn=norm, padding=1, indice_key='subm' + level))
                else:
                    built.append(SparseBasicBlock(spec[1], spec[1], norm_fn=norm, indice_key='res' + level))
            setattr(self, name, spconv.SparseSequential(*built))
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(out_in_channels, 128, (3, 1, 1), stride=(2, 1, 1), padding=model_cfg.get('last_pad', 0), bias=False,
                                indice_key='spconv_down2'),
            norm(128), nn.ReLU())
        self.num_point_features = 128

    def forward(self, batch_dict):
        """batch_dict: batch_size, voxel_features (N, C), voxel_coords (N, 4) [b, z, y, x] (float or int)."""
        x = spconv.SparseConvTensor(features=batch_dict['voxel_features'], indices=batch_dict['voxel_coords'].int(),
                                    spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        stages = [getattr(self, name) for name in _STAGE_ORDER]
        if not torch.is_grad_enabled() and not self.training and x.features.is_cuda and x.indices.shape[0] > 0:
            # inference (eval mode: one fused launch per layer, the host is far ahead of the GPU): the next stage's rulebooks
            # are built on a second stream under this stage's GEMMs.  With train-mode BatchNorm under no_grad the feature pass
            # is launch-bound and the interleaving delays the rulebook chain (6.4 vs 5.5 ms), so that case keeps the prebuilt pass.
            cap = int(getattr(self, 'graph_capacity', 0) or 0)
            if cap >= x.indices.shape[0]:
                # opt-in (not in the reference): the whole forward replayed as one captured hipGraph over inputs padded to
                # `graph_capacity` rows (spconv.GraphedStages); outputs are exact but live in the graph's memory until the next call
                key = (x.batch_size, cap, x.features.shape[1], str(x.features.device))
                graphs = self.__dict__.setdefault('_graphs', {})
                if key not in graphs:
                    graphs[key] = spconv.GraphedStages(stages, x.spatial_shape, x.batch_size, x.features.shape[1], cap, x.features.device)
                taps = dict(zip(_STAGE_ORDER, graphs[key](x.features, x.indices)))
            else:
                taps = dict(zip(_STAGE_ORDER, spconv.run_stages_pipelined(stages, x)))
            x = taps[_STAGE_ORDER[-1]]
        else:
            spconv.prebuild_rulebooks(stages, x.indices.contiguous(), x.spatial_shape, x.batch_size, x.indice_dict)
            taps = {}
            for name, stage in zip(_STAGE_ORDER, stages):
                x = stage(x)
                taps[name] = x
        batch_dict.update({'encoded_spconv_tensor': x, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': {'x_conv%d' % i: taps['conv%d' % i] for i in (1, 2, 3, 4)}})
        return batch_dict


class VoxelBackBone8x(_VoxelBackBoneBase):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self._assemble(model_cfg, input_channels, grid_size, _PLAIN_STAGES, out_in_channels=64)


class VoxelResBackBone8x(_VoxelBackBoneBase):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self._assemble(model_cfg, input_channels, grid_size, _RES_STAGES, out_in_channels=128)
