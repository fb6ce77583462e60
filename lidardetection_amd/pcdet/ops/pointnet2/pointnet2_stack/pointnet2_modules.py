"""Stacked-batch set abstraction / feature propagation modules with the reference's class names, keyword arguments and
parameter layout (pcdet/ops/pointnet2/pointnet2_stack/pointnet2_modules.py: StackSAModuleMSG :10-92, StackPointnetFPModule
:95-137), so its checkpoints load."""
from typing import List

import torch
import torch.nn as nn

from . import pointnet2_utils
from .. import _common as C


class StackSAModuleMSG(nn.Module):
    def __init__(self, *, radii: List[float], nsamples: List[int], mlps: List[List[int]], use_xyz: bool = True,
                 pool_method='max_pool'):
        super().__init__()
        if not (len(radii) == len(nsamples) == len(mlps)):
            raise AssertionError('one radius, sample count and MLP per scale')
        self.groupers, self.mlps = nn.ModuleList(), nn.ModuleList()
        for radius, nsample, widths in zip(radii, nsamples, mlps):
            self.groupers.append(pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz))
            if use_xyz:
                widths[0] += 3        # modifies the caller's list, as the reference does (callers read it back)
            self.mlps.append(C.shared_mlp(widths))
        self.pool_method = pool_method
        self.init_weights()

    def init_weights(self):
        for layer in self.modules():
            if isinstance(layer, nn.Conv2d):
                nn.init.kaiming_normal_(layer.weight)
                if layer.bias is not None:
                    nn.init.zeros_(layer.bias)
            elif isinstance(layer, nn.BatchNorm2d):
                nn.init.ones_(layer.weight)
                nn.init.zeros_(layer.bias)

    def _folded_layers(self, k):
        """scale k's shared MLP as row-major GEMM operands: [(W (Cin rounded up to 4, Cout), shift (Cout))] with the eval-mode
        BatchNorm folded in; cached until a parameter changes"""
        mods = list(self.mlps[k])
        srcs = [t for m in mods for t in (getattr(m, 'weight', None), getattr(m, 'bias', None), getattr(m, 'running_mean', None),
                                          getattr(m, 'running_var', None)) if t is not None]
        key = tuple((t.data_ptr(), t._version) for t in srcs)
        cache = self.__dict__.setdefault('_fold_cache', {})
        if k not in cache or cache[k][0] != key:
            layers = []
            with torch.no_grad():
                for conv, bn in zip(mods[0::3], mods[1::3]):
                    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    w = conv.weight[:, :, 0, 0].t() * scale.view(1, -1)                       # (Cin, Cout)
                    shift = bn.bias - bn.running_mean * scale
                    if conv.bias is not None:
                        shift = shift + conv.bias * scale
                    pad = (-w.shape[0]) % 4 if not layers else 0       # first layer only: the gathered rows' pitch
                    if pad:
                        w = torch.cat((w, w.new_zeros(pad, w.shape[1])), dim=0)
                    layers.append((w.contiguous(), shift.contiguous()))
            cache[k] = (key, layers)
        return cache[k][1]

    def _inference_ready(self, xyz):
        if torch.is_grad_enabled() or self.training or self.pool_method != 'max_pool' or not xyz.is_cuda:
            return False
        for mlp in self.mlps:
            mods = list(mlp)
            if len(mods) % 3 or not all(isinstance(c, nn.Conv2d) and isinstance(b, nn.BatchNorm2d) and isinstance(a, nn.ReLU)
                                        and not b.training and b.track_running_stats
                                        for c, b, a in zip(mods[0::3], mods[1::3], mods[2::3])):
                return False
        return True

    def forward_inference(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None):
        """The same result as forward() in eval mode, built for the GPU: groups are gathered ROW-major ((M * nsample, C) — one
        contiguous run per neighbour, lidar_group_rows_stack), the shared MLP is a chain of plain GEMMs with BatchNorm folded
        into the weights and the shift + ReLU applied in place, and the max runs over contiguous blocks of nsample rows.  The
        (M, C, nsample) tensor of the reference layout and the 1x1 convolution over its strided view never exist."""
        from .....ext import pointnet2_stack_cuda as native
        C.require_contiguous(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt)
        n_batch, n_query = xyz_batch_cnt.shape[0], new_xyz.shape[0]
        feats = None if features is None else features.contiguous()
        width = 0 if feats is None else feats.shape[1]
        per_scale = []
        idxs = [C.zeros_i32((n_query, g.nsample), xyz.device) for g in self.groupers]
        ga = self.groupers
        # scales in pairs (one pass over the distances for both radii), through a cell grid over the candidates when there are
        # enough of them per batch element to pay for the binning pass
        use_grid = xyz.shape[0] >= native.GRID_MIN_POINTS * n_batch and max(g.nsample for g in ga) <= 64
        for k in range(0, len(ga) - 1, 2):
            if use_grid:
                native.ball_query_grid_wrapper(n_batch, n_query, ga[k].radius, ga[k].nsample, ga[k + 1].radius, ga[k + 1].nsample,
                                               new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idxs[k], idxs[k + 1])
            else:
                native.ball_query2_wrapper(n_batch, n_query, ga[k].radius, ga[k].nsample, ga[k + 1].radius, ga[k + 1].nsample,
                                           new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, idxs[k], idxs[k + 1])
        if len(ga) % 2:
            if use_grid:
                native.ball_query_grid_wrapper(n_batch, n_query, ga[-1].radius, ga[-1].nsample, None, None, new_xyz,
                                               new_xyz_batch_cnt, xyz, xyz_batch_cnt, idxs[-1], None)
            else:
                native.ball_query_wrapper(n_batch, n_query, ga[-1].radius, ga[-1].nsample, new_xyz, new_xyz_batch_cnt, xyz,
                                          xyz_batch_cnt, idxs[-1])
        src = None
        for k, grouper in enumerate(self.groupers):
            layers = self._folded_layers(k)
            idx = idxs[k]
            w1, b1 = layers[0]
            if w1.shape[1] % 4 == 0:
                # layer 1 in front of the gather (it commutes with it): one GEMM row per SOURCE point instead of one per
                # (query, sample) pair; the layer is then a gather of its output rows, a per-query term and the ReLU
                if grouper.use_xyz:
                    if src is None:
                        src = xyz if feats is None else torch.cat((xyz, feats), dim=1)
                    table = torch.addmm(b1, src, w1[:src.shape[1]])
                    query_term = torch.mm(new_xyz, w1[:3])
                else:
                    table, query_term = torch.addmm(b1, feats, w1[:width]), None
                if len(layers) == 2 and native.sa_layer2_max_supported(w1.shape[1], layers[1][0].shape[1], grouper.nsample):
                    # two-layer scale (every one in PV-RCNN): gather + second layer on the matrix cores + max in ONE kernel
                    pooled = C.empty_f32((n_query, layers[1][0].shape[1]), xyz.device)
                    native.sa_layer2_max_wrapper(n_batch, n_query, grouper.nsample, table, query_term, torch.relu(b1), layers[1][0],
                                                 layers[1][1], xyz_batch_cnt, idx, new_xyz_batch_cnt, pooled)
                    per_scale.append(pooled)
                    continue
                rows = C.empty_f32((n_query * grouper.nsample, w1.shape[1]), xyz.device)
                native.group_rows_affine_wrapper(n_batch, n_query, w1.shape[1], grouper.nsample, table, query_term,
                                                 torch.relu(b1), xyz_batch_cnt, idx, new_xyz_batch_cnt, rows)
                rest = layers[1:]
            else:
                stride = w1.shape[0]
                rows = C.empty_f32((n_query * grouper.nsample, stride), xyz.device)
                native.group_rows_wrapper(n_batch, n_query, width, grouper.nsample, grouper.use_xyz, stride, xyz, new_xyz, feats,
                                          xyz_batch_cnt, idx, new_xyz_batch_cnt, rows)
                rest = layers
            for w, shift in rest:
                rows = C.addmm_act(shift, rows, w)
            per_scale.append(rows.view(n_query, grouper.nsample, -1).amax(dim=1))
        return new_xyz, torch.cat(per_scale, dim=1)

    def forward(self, xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features=None, empty_voxel_set_zeros=True):
        """xyz (N, 3), features (N, C), new_xyz (M, 3) -> (new_xyz, (M, sum of the scales' last MLP widths))"""
        if self._inference_ready(xyz) and new_xyz.shape[0] > 0:
            return self.forward_inference(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)
        per_scale = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            grouped, _ = grouper(xyz, xyz_batch_cnt, new_xyz, new_xyz_batch_cnt, features)      # (M, C, nsample)
            y = mlp(grouped.permute(1, 0, 2).unsqueeze(0))                                      # (1, C', M, nsample)
            per_scale.append(C.pool_over_samples(y, self.pool_method)[0].t())                    # (M, C')
        return new_xyz, torch.cat(per_scale, dim=1)


class StackPointnetFPModule(nn.Module):
    def __init__(self, *, mlp: List[int]):
        super().__init__()
        self.mlp = C.shared_mlp(mlp)

    def forward(self, unknown, unknown_batch_cnt, known, known_batch_cnt, unknown_feats=None, known_feats=None):
        """known_feats (M, C2) interpolated onto unknown (N, 3) by inverse distance to the 3 nearest known points,
        concatenated with unknown_feats, then the shared MLP -> (N, C_out)"""
        dist, idx = pointnet2_utils.three_nn(unknown, unknown_batch_cnt, known, known_batch_cnt)
        y = pointnet2_utils.three_interpolate(known_feats, idx, C.inverse_distance_weights(dist))
        if unknown_feats is not None:
            y = torch.cat((y, unknown_feats), dim=1)
        return self.mlp(y.t()[None, :, :, None])[0, :, :, 0].t()
