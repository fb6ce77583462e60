"""fp64 CPU replay of the stacked set-abstraction modules — TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Restates, on top of the C oracle's ball query / grouping (oracle/src/points_oracle.c, which follows
/root/reference/pcdet/ops/pointnet2/pointnet2_stack/src/ball_query_gpu.cu:16-66 and group_points_gpu.cu:71-102), what
QueryAndGroup.forward (pcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:119-155) and StackSAModuleMSG.forward
(pointnet2_modules.py:58-92) compute: first-nsample ball query, -1 sentinel -> empty-ball mask, grouped xyz minus the centre,
empty balls zeroed, [xyz | features] concatenated, shared 1x1-conv MLP with eval-mode BatchNorm, max over the samples.  The MLP
runs in float64 with the module's own weights.  PARITY UNPINNED beyond the C oracle's own pins (the reference has no test here).
"""
import numpy as np

from . import c_oracle


def query_and_group(radius, nsample, xyz, xyz_cnt, new_xyz, new_cnt, features, use_xyz=True):
    """-> (grouped (M, 3 + C, nsample) float32 exactly as the reference builds it, idx (M, nsample), empty mask (M,))"""
    xyz, new_xyz = np.ascontiguousarray(xyz, np.float32), np.ascontiguousarray(new_xyz, np.float32)
    idx = c_oracle.ball_query_stack(radius, nsample, xyz, xyz_cnt, new_xyz, new_cnt)
    empty = idx[:, 0] == -1
    idx[empty] = 0
    gx = c_oracle.group_points_stack(xyz, xyz_cnt, idx, new_cnt) - new_xyz[:, :, None]      # float32, like the device
    gx[empty] = 0
    if features is None:
        return gx, idx, empty
    gf = c_oracle.group_points_stack(np.ascontiguousarray(features, np.float32), xyz_cnt, idx, new_cnt)
    gf[empty] = 0
    return (np.concatenate([gx, gf], 1) if use_xyz else gf), idx, empty


def shared_mlp_fp64(mlp, x):
    """mlp = torch Sequential of (Conv2d 1x1 no bias, BatchNorm2d eval, ReLU) triplets; x (M, C, S) -> (M, C', S) float64"""
    import torch.nn as nn
    x = np.asarray(x, np.float64)
    mods = list(mlp)
    for conv, bn, act in zip(mods[0::3], mods[1::3], mods[2::3]):
        assert isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and isinstance(act, nn.ReLU) and not bn.training
        w = conv.weight.detach().cpu().double().numpy()[:, :, 0, 0]                         # (Cout, Cin)
        x = np.einsum("oc,mcs->mos", w, x)
        g = lambda t: t.detach().cpu().double().numpy()[None, :, None]
        x = (x - g(bn.running_mean)) / np.sqrt(g(bn.running_var) + bn.eps) * g(bn.weight) + g(bn.bias)
        x = np.maximum(x, 0.0)
    return x


def stack_sa_msg(module, xyz, xyz_cnt, new_xyz, new_cnt, features):
    """StackSAModuleMSG.forward (max_pool) -> (M, sum of the scales' last widths) float64"""
    outs = []
    for grouper, mlp in zip(module.groupers, module.mlps):
        g, _, _ = query_and_group(grouper.radius, grouper.nsample, xyz, xyz_cnt, new_xyz, new_cnt, features, grouper.use_xyz)
        outs.append(shared_mlp_fp64(mlp, g).max(axis=-1))
    return np.concatenate(outs, 1)
