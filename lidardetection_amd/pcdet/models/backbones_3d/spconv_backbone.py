"""Mirror of pcdet/models/backbones_3d/spconv_backbone.py: post_act_block (:7-26), SparseBasicBlock (:29-65),
VoxelBackBone8x (:68-163), VoxelResBackBone8x (:166-261) — built on lidardetection_amd.spconv.  Module / parameter
names are the reference's, so its checkpoints (weight layout (kD,kH,kW,Cin,Cout)) load unchanged."""
from functools import partial

import torch.nn as nn

from .... import spconv


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm', norm_fn=None):
    if conv_type == 'subm':
        conv = spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = spconv.SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                                   indice_key=indice_key)
    elif conv_type == 'inverseconv':
        conv = spconv.SparseInverseConv3d(in_channels, out_channels, kernel_size, indice_key=indice_key, bias=False)
    else:
        raise NotImplementedError
    return spconv.SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


class SparseBasicBlock(spconv.SparseModule):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        bias = norm_fn is not None
        self.conv1 = spconv.SubMConv3d(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=bias, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = spconv.SubMConv3d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=bias, indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        import torch
        if not torch.is_grad_enabled() and not self.bn1.training and x.indices.shape[0] != 0:
            # inference: each conv + BN (+ residual) + ReLU is one launch (spconv.conv.forward_fused)
            identity = x if self.downsample is None else self.downsample(x)
            out = self.conv1.forward_fused(x, self.bn1, relu=True)
            return self.conv2.forward_fused(out, self.bn2, relu=True, residual=identity.features)
        identity = x
        out = self.conv1(x)
        out.features = self.relu(self.bn1(out.features))
        out = self.conv2(out)
        out.features = self.bn2(out.features)
        if self.downsample is not None:
            identity = self.downsample(x)
        out.features = self.relu(out.features + identity.features)
        return out


class _VoxelBackBoneBase(nn.Module):
    def _finish(self, batch_dict, x_convs, out):
        batch_dict.update({'encoded_spconv_tensor': out, 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': dict(zip(('x_conv1', 'x_conv2', 'x_conv3', 'x_conv4'), x_convs))})
        return batch_dict

    def forward(self, batch_dict):
        """batch_dict: batch_size, voxel_features (N, C), voxel_coords (N, 4) [b, z, y, x] (float or int)."""
        sp = spconv.SparseConvTensor(features=batch_dict['voxel_features'], indices=batch_dict['voxel_coords'].int(),
                                     spatial_shape=self.sparse_shape, batch_size=batch_dict['batch_size'])
        # all rulebooks first (coordinates only; the 4 strided convs each read one int back), then a sync-free feature pass
        stages = [getattr(self, name) for name in ('conv_input', 'conv1', 'conv2', 'conv3', 'conv4', 'conv_out')]
        spconv.prebuild_rulebooks(stages, sp.indices.contiguous(), sp.spatial_shape, sp.batch_size, sp.indice_dict)
        x = self.conv_input(sp)
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        return self._finish(batch_dict, (x1, x2, x3, x4), self.conv_out(x4))


class VoxelBackBone8x(_VoxelBackBoneBase):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (list(grid_size[::-1]))]
        self.sparse_shape[0] += 1                                   # grid_size[::-1] + [1, 0, 0]
        block = post_act_block
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        self.conv1 = spconv.SparseSequential(block(16, 16, 3, norm_fn=norm_fn, padding=1, indice_key='subm1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'),
            block(32, 32, 3, norm_fn=norm_fn, padding=1, indice_key='subm2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 64, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'),
            block(64, 64, 3, norm_fn=norm_fn, padding=1, indice_key='subm4'))
        last_pad = self.model_cfg.get('last_pad', 0)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(64, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False, indice_key='spconv_down2'),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128


class VoxelResBackBone8x(_VoxelBackBoneBase):
    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.sparse_shape = [int(v) for v in (list(grid_size[::-1]))]
        self.sparse_shape[0] += 1
        block = post_act_block
        self.conv_input = spconv.SparseSequential(
            spconv.SubMConv3d(input_channels, 16, 3, padding=1, bias=False, indice_key='subm1'), norm_fn(16), nn.ReLU())
        self.conv1 = spconv.SparseSequential(SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key='res1'),
                                             SparseBasicBlock(16, 16, norm_fn=norm_fn, indice_key='res1'))
        self.conv2 = spconv.SparseSequential(
            block(16, 32, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res2'), SparseBasicBlock(32, 32, norm_fn=norm_fn, indice_key='res2'))
        self.conv3 = spconv.SparseSequential(
            block(32, 64, 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res3'), SparseBasicBlock(64, 64, norm_fn=norm_fn, indice_key='res3'))
        self.conv4 = spconv.SparseSequential(
            block(64, 128, 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res4'), SparseBasicBlock(128, 128, norm_fn=norm_fn, indice_key='res4'))
        last_pad = self.model_cfg.get('last_pad', 0)
        self.conv_out = spconv.SparseSequential(
            spconv.SparseConv3d(128, 128, (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False, indice_key='spconv_down2'),
            norm_fn(128), nn.ReLU())
        self.num_point_features = 128
