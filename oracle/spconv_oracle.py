"""CPU oracle for the sparse-convolution path — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

spconv (v1.0@8da6f96 / v1.2, docs/INSTALL.md:9,28-29) is an un-vendored third-party dependency that is absent from
/root/reference and from this image, and the reference holds no test or golden vector at this boundary:
PARITY UNPINNED w.r.t. upstream.  The oracle restates the published semantics two ways:
  * rulebook  : brute-force enumeration of (offset k, input coord, output coord) triples in pure Python/numpy;
  * features  : dense torch.nn.functional.conv3d / conv_transpose3d on the densified input, sampled at the active
                output sites (cross-correlation, weight (kD,kH,kW,Cin,Cout) -> (Cout,Cin,kD,kH,kW)).
Call sites that fix the shapes: /root/reference/pcdet/models/backbones_3d/spconv_backbone.py:76-116.
"""
import numpy as np
import torch
import torch.nn.functional as F


def out_shape(shape, ksize, stride, padding):
    return [(i + 2 * p - k) // s + 1 for i, k, s, p in zip(shape, ksize, stride, padding)]


def rulebook(indices, shape, ksize, stride, padding, subm):
    """-> (set of (k, in_coord(b,z,y,x), out_coord(b,z,y,x)), sorted list of active output coords)."""
    idx = [tuple(int(v) for v in r) for r in np.asarray(indices)]
    active = set(idx)
    triples = set()
    if subm:
        c = [k // 2 for k in ksize]
        for (b, z, y, x) in idx:            # output site j == input site; neighbour = site + (k - centre)
            for kz in range(ksize[0]):
                for ky in range(ksize[1]):
                    for kx in range(ksize[2]):
                        q = (b, z + kz - c[0], y + ky - c[1], x + kx - c[2])
                        if q in active:
                            triples.add(((kz * ksize[1] + ky) * ksize[2] + kx, q, (b, z, y, x)))
        return triples, sorted(active)
    osz = out_shape(shape, ksize, stride, padding)
    outs = set()
    for (b, z, y, x) in idx:
        for kz in range(ksize[0]):
            for ky in range(ksize[1]):
                for kx in range(ksize[2]):
                    t = (z + padding[0] - kz, y + padding[1] - ky, x + padding[2] - kx)
                    if all(v >= 0 and v % s == 0 for v, s in zip(t, stride)):
                        o = tuple(v // s for v, s in zip(t, stride))
                        if all(v < m for v, m in zip(o, osz)):
                            triples.add(((kz * ksize[1] + ky) * ksize[2] + kx, (b, z, y, x), (b,) + o))
                            outs.add((b,) + o)
    return triples, sorted(outs)


def densify(features, indices, batch, shape):
    C = features.shape[1]
    d = torch.zeros((batch, C, *shape), dtype=torch.float64)
    idx = torch.as_tensor(np.asarray(indices)).long()
    d[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = torch.as_tensor(features).double()
    return d


def _w(weight):
    return torch.as_tensor(weight).double().permute(4, 3, 0, 1, 2).contiguous()    # (Cout, Cin, kD, kH, kW)


def conv_features(features, indices, batch, shape, weight, bias, ksize, stride, padding, subm, out_coords):
    """Dense-conv oracle sampled at out_coords ((M,4) array) -> (M, Cout) float64."""
    d = densify(features, indices, batch, shape)
    if subm:
        o = F.conv3d(d, _w(weight), None, stride=1, padding=[k // 2 for k in ksize])
    else:
        o = F.conv3d(d, _w(weight), None, stride=stride, padding=padding)
    oc = torch.as_tensor(np.asarray(out_coords)).long()
    r = o[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]]
    return r + torch.as_tensor(bias).double() if bias is not None else r


def inverse_conv_features(features, indices, batch, shape_small, weight, bias, ksize, stride, padding, orig_coords, orig_shape):
    """SparseInverseConv3d: transposed conv of the (small-grid) input, sampled at the paired conv's input sites."""
    d = densify(features, indices, batch, shape_small)
    wt = torch.as_tensor(weight).double().permute(3, 4, 0, 1, 2).contiguous()      # (Cin, Cout, kD, kH, kW)
    base = [(o - 1) * s - 2 * p + k for o, s, p, k in zip(shape_small, stride, padding, ksize)]
    opad = [max(0, t - b) for t, b in zip(orig_shape, base)]
    o = F.conv_transpose3d(d, wt, None, stride=stride, padding=padding, output_padding=opad)
    oc = torch.as_tensor(np.asarray(orig_coords)).long()
    r = o[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]]
    return r + torch.as_tensor(bias).double() if bias is not None else r
