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


def _idw(dist):
    """inverse-distance weights of the 3 neighbours (pointnet2_modules.py(stack):123-126 / (batch):150-153), float32 like the device"""
    inv = np.float32(1.0) / (np.asarray(dist, np.float32) + np.float32(1e-8))
    return (inv / inv.sum(-1, keepdims=True)).astype(np.float32)


def stack_fp(module, unknown, unknown_cnt, known, known_cnt, unknown_feats, known_feats):
    """StackPointnetFPModule.forward (pointnet2_modules.py(stack):111-137) -> (N, mlp[-1]) float64"""
    d2, idx = c_oracle.three_nn_stack(unknown, unknown_cnt, known, known_cnt)
    w = _idw(np.sqrt(d2))
    y = c_oracle.three_interpolate_stack(np.ascontiguousarray(known_feats, np.float32), idx, w).astype(np.float64)
    if unknown_feats is not None:
        y = np.concatenate([y, np.asarray(unknown_feats, np.float64)], 1)
    return shared_mlp_fp64(module.mlp, y.T[None])[0].T                    # (1, C, N): one 'ball' whose samples are the points


def batch_sa_msg(module, xyz, features):
    """PointnetSAModuleMSG.forward (pointnet2_batch/pointnet2_modules.py:28-58, :61-101), dense batches:
    xyz (B, N, 3), features (B, C, N) -> (new_xyz (B, npoint, 3) float32, (B, sum widths, npoint) float64)"""
    xyz = np.ascontiguousarray(xyz, np.float32)
    B = xyz.shape[0]
    picked = c_oracle.fps(xyz, module.npoint)
    new_xyz = np.stack([xyz[b][picked[b]] for b in range(B)], 0)
    outs = []
    for grouper, mlp in zip(module.groupers, module.mlps):
        idx = c_oracle.ball_query_batch(grouper.radius, grouper.nsample, xyz, new_xyz)                 # no -1 marker in this variant
        g = c_oracle.group_points_batch(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - new_xyz.transpose(0, 2, 1)[..., None]
        if features is not None:
            gf = c_oracle.group_points_batch(np.ascontiguousarray(features, np.float32), idx)
            g = np.concatenate([g, gf], 1) if grouper.use_xyz else gf
        y = np.stack([shared_mlp_fp64(mlp, g[b].transpose(1, 0, 2)).max(-1).T for b in range(B)], 0)   # (B, C', npoint)
        outs.append(y)
    return new_xyz, np.concatenate(outs, 1)


def batch_fp(module, unknown, known, unknown_feats, known_feats):
    """PointnetFPModule.forward (pointnet2_batch/pointnet2_modules.py:124-170) -> (B, mlp[-1], n) float64"""
    d2, idx = c_oracle.three_nn_batch(unknown, known)
    w = _idw(np.sqrt(d2))
    y = c_oracle.three_interpolate_batch(np.ascontiguousarray(known_feats, np.float32), idx, w).astype(np.float64)   # (B, C2, n)
    if unknown_feats is not None:
        y = np.concatenate([y, np.asarray(unknown_feats, np.float64)], 1)
    return np.stack([shared_mlp_fp64(module.mlp, y[b][None])[0] for b in range(y.shape[0])], 0)
