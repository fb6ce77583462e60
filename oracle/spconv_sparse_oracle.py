"""Sparse fp64 CPU oracle for the sparse-convolution path at FULL grid sizes — TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

oracle/spconv_oracle.py densifies the grid and runs torch's conv3d, which cannot hold a 41 x 1600 x 1408 SECOND grid.  This
file restates the same semantics (out[o] = sum_k in[o * s - p + k] @ W[k] over ACTIVE inputs;
SubM: outputs = inputs; regular: outputs = every site reached by at least one active input) without ever leaving the active
set: coordinates are linearised to int64 keys, neighbours are found by binary search in the sorted key array, and every kernel
offset contributes one float64 matmul of its gathered rows.  PARITY UNPINNED w.r.t. upstream spconv (absent, un-vendored;
docs/INSTALL.md:9,28-29) exactly as spconv_oracle.py; the two oracles are checked against each other on small grids
(tests/test_oracle_pins.py).  Call sites that fix the shapes: /root/reference/pcdet/models/backbones_3d/spconv_backbone.py:76-116.
"""
import numpy as np


def out_shape(shape, ksize, stride, padding):
    return [(i + 2 * p - k) // s + 1 for i, k, s, p in zip(shape, ksize, stride, padding)]


def _keys(idx, shape):
    D, H, W = (int(v) for v in shape)
    i = np.asarray(idx, np.int64)
    return ((i[:, 0] * D + i[:, 1]) * H + i[:, 2]) * W + i[:, 3]


def _offsets(ksize):
    return [(kz, ky, kx) for kz in range(ksize[0]) for ky in range(ksize[1]) for kx in range(ksize[2])]


def subm_conv(feats, idx, shape, weight, bias, ksize):
    """SubMConv3d: feats (N, Cin), idx (N, 4) [b, z, y, x] (any order, unique), weight (kD, kH, kW, Cin, Cout) -> (N, Cout) f64
    in the input's row order, plus the (k, in_row, out_row) pair count per offset."""
    feats = np.asarray(feats, np.float64)
    idx = np.asarray(idx, np.int64)
    w = np.asarray(weight, np.float64).reshape(-1, weight.shape[-2], weight.shape[-1])
    keys = _keys(idx, shape)
    order = np.argsort(keys, kind="stable")
    skeys = keys[order]
    out = np.zeros((idx.shape[0], w.shape[2]), np.float64)
    centre = [k // 2 for k in ksize]
    lim = np.asarray(shape, np.int64)
    n_k = []
    for k, off in enumerate(_offsets(ksize)):
        q = idx.copy()
        q[:, 1:] += np.asarray(off, np.int64) - np.asarray(centre, np.int64)           # neighbour = site + (k - centre)
        ok = np.all((q[:, 1:] >= 0) & (q[:, 1:] < lim), axis=1)
        qk = _keys(q[ok], shape)
        pos = np.searchsorted(skeys, qk)
        pos[pos >= skeys.size] = 0
        hit = skeys[pos] == qk
        rows_out = np.nonzero(ok)[0][hit]
        rows_in = order[pos[hit]]
        n_k.append(int(rows_out.size))
        if rows_out.size:
            out[rows_out] += feats[rows_in] @ w[k]
    if bias is not None:
        out += np.asarray(bias, np.float64)
    return out, n_k


def sparse_conv(feats, idx, shape, weight, bias, ksize, stride, padding):
    """SparseConv3d -> (out_feats (M, Cout) f64, out_idx (M, 4) ascending (b, z, y, x), out_shape, pairs per offset)."""
    feats = np.asarray(feats, np.float64)
    idx = np.asarray(idx, np.int64)
    w = np.asarray(weight, np.float64).reshape(-1, weight.shape[-2], weight.shape[-1])
    osz = out_shape(shape, ksize, stride, padding)
    s = np.asarray(stride, np.int64)
    per_k = []
    for off in _offsets(ksize):
        t = idx[:, 1:] + np.asarray(padding, np.int64) - np.asarray(off, np.int64)       # o = (in + p - k) / s
        ok = np.all((t >= 0) & (t % s == 0), axis=1)
        o = t // s
        ok &= np.all(o < np.asarray(osz, np.int64), axis=1)
        rows_in = np.nonzero(ok)[0]
        oc = np.concatenate([idx[rows_in, :1], o[rows_in]], axis=1)
        per_k.append((rows_in, _keys(oc, osz), oc))
    all_keys = np.concatenate([p[1] for p in per_k])
    all_oc = np.concatenate([p[2] for p in per_k])
    ukeys, first = np.unique(all_keys, return_index=True)
    out_idx = all_oc[first]
    out = np.zeros((ukeys.size, w.shape[2]), np.float64)
    n_k = []
    for k, (rows_in, okeys, _) in enumerate(per_k):
        n_k.append(int(rows_in.size))
        if rows_in.size:
            out[np.searchsorted(ukeys, okeys)] += feats[rows_in] @ w[k]                     # one output per input within an offset
    if bias is not None:
        out += np.asarray(bias, np.float64)
    return out, out_idx, osz, n_k


def inverse_conv(feats_small, idx_small, shape_small, idx_orig, shape_orig, weight, bias, ksize, stride, padding):
    """SparseInverseConv3d (Appendix A.3; reference call sites pcdet/models/backbones_3d/spconv_unet.py:113-123): the paired
    SparseConv3d's rulebook with inputs and outputs swapped — for every pair (k, i -> o) of that convolution (o = (i + p - k) / s
    integral and in bounds), out[i] += feats_small[o] @ W[k]; outputs are the paired convolution's INPUT sites idx_orig (N, 4), in
    that row order.  feats_small (M, Cin) follows idx_small (M, 4) — any order; weight (kD, kH, kW, Cin, Cout).  -> (N, Cout) f64."""
    feats_small = np.asarray(feats_small, np.float64)
    idx_orig = np.asarray(idx_orig, np.int64)
    w = np.asarray(weight, np.float64).reshape(-1, weight.shape[-2], weight.shape[-1])
    skeys = _keys(idx_small, shape_small)
    order = np.argsort(skeys, kind="stable")
    skeys = skeys[order]
    s = np.asarray(stride, np.int64)
    out = np.zeros((idx_orig.shape[0], w.shape[2]), np.float64)
    for k, off in enumerate(_offsets(ksize)):
        t = idx_orig[:, 1:] + np.asarray(padding, np.int64) - np.asarray(off, np.int64)
        ok = np.all((t >= 0) & (t % s == 0), axis=1)
        o = t // s
        ok &= np.all(o < np.asarray(shape_small, np.int64), axis=1)
        rows_i = np.nonzero(ok)[0]
        okeys = _keys(np.concatenate([idx_orig[rows_i, :1], o[rows_i]], axis=1), shape_small)
        pos = np.searchsorted(skeys, okeys)
        pos[pos >= skeys.size] = 0
        hit = skeys[pos] == okeys                       # (every such o IS an output site of the paired conv; kept for safety)
        if hit.any():
            out[rows_i[hit]] += feats_small[order[pos[hit]]] @ w[k]
    if bias is not None:
        out += np.asarray(bias, np.float64)
    return out


def batchnorm_eval(x, weight, bias, mean, var, eps):
    return (x - np.asarray(mean, np.float64)) / np.sqrt(np.asarray(var, np.float64) + eps) * np.asarray(weight, np.float64) \
        + np.asarray(bias, np.float64)
