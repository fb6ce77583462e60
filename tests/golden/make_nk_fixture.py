"""Emits tests/golden/spconv_nk_second_kitti_bs16.json — SURVEY.md 8(d): "FLOPs = 2 * sum_layers sum_k n_k * Cin * Cout with n_k taken from the
oracle rulebook of the fixed synthetic frame (commit the per-layer n_k table as a fixture)".

Fixed synthetic batch = what bench.py's `extra.spconv_gemm` and tools/sorted_gemm_bench.py run: synth.cloud_ring(2000 .. 2015), SECOND-KITTI
voxel grid (0.05 x 0.05 x 0.1 m, 41 x 1600 x 1408, <= 5 points / voxel, <= 16 000 voxels per frame), voxelised by the sequential C oracle.
The rulebooks are the sparse fp64 oracle's (oracle/spconv_sparse_oracle.py: sorted keys + binary search), walked through the layer
geometry of VoxelBackBone8x (/root/reference/pcdet/models/backbones_3d/spconv_backbone.py:76-116).  No GPU, no reference code is run.
usage: python tests/golden/make_nk_fixture.py   (about half a minute)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lidardetection_amd import synth  # noqa: E402
from oracle import c_oracle, spconv_sparse_oracle as sp  # noqa: E402

# (name, kind, Cin, Cout, ksize, stride, padding) in execution order: spconv_backbone.py:76-116
LAYERS = [("conv_input", "subm", 4, 16, (3, 3, 3), None, None), ("conv1.0", "subm", 16, 16, (3, 3, 3), None, None),
          ("conv2.0", "conv", 16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), ("conv2.1", "subm", 32, 32, (3, 3, 3), None, None),
          ("conv2.2", "subm", 32, 32, (3, 3, 3), None, None), ("conv3.0", "conv", 32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
          ("conv3.1", "subm", 64, 64, (3, 3, 3), None, None), ("conv3.2", "subm", 64, 64, (3, 3, 3), None, None),
          ("conv4.0", "conv", 64, 64, (3, 3, 3), (2, 2, 2), (0, 1, 1)), ("conv4.1", "subm", 64, 64, (3, 3, 3), None, None),
          ("conv4.2", "subm", 64, 64, (3, 3, 3), None, None), ("conv_out", "conv", 64, 128, (3, 1, 1), (2, 1, 1), (0, 0, 0))]


def batch_coords(B=16):
    idx = []
    for f in range(B):
        _, coords, _ = c_oracle.voxelize(synth.cloud_ring(2000 + f), synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000)
        idx.append(np.concatenate([np.full((len(coords), 1), f, np.int64), coords.astype(np.int64)], 1))
    return np.concatenate(idx, 0)


def layer_table(idx, shape):
    rows = []
    for name, kind, cin, cout, ks, st, pd in LAYERS:
        f = np.zeros((idx.shape[0], 1))
        w = np.zeros(tuple(ks) + (1, 1))
        if kind == "subm":
            _, n_k = sp.subm_conv(f, idx, shape, w, None, ks)
            n_out = idx.shape[0]
        else:
            _, idx, shape, n_k = sp.sparse_conv(f, idx, shape, w, None, ks, st, pd)
            n_out = idx.shape[0]
        rows.append({"layer": name, "kind": kind, "cin": cin, "cout": cout, "rows_out": int(n_out), "n_k": [int(v) for v in n_k],
                     "pairs": int(sum(n_k)), "gflop": 2.0 * sum(n_k) * cin * cout / 1e9})
    return rows


def build():
    idx = batch_coords()
    rows = layer_table(idx, [41, 1600, 1408])
    return {"what": "per-layer, per-offset pair counts n_k of VoxelBackBone8x on cloud_ring(2000..2015), SECOND-KITTI grid, bs 16 (SURVEY 8d)",
            "voxels": int(idx.shape[0]), "layers": rows, "gflop_useful_total": float(sum(r["gflop"] for r in rows)),
            "gflop_useful_gemm_kernels": float(sum(r["gflop"] for r in rows))}


if __name__ == "__main__":
    out = build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spconv_nk_second_kitti_bs16.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(path, out["voxels"], "voxels", f'{out["gflop_useful_total"]:.3f} GFLOP useful')
    for r in out["layers"]:
        print(f'  {r["layer"]:11s} {r["cin"]:3d}->{r["cout"]:3d} rows {r["rows_out"]:7d} pairs {r["pairs"]:9d} {r["gflop"]:8.3f} GFLOP')
