"""Pins the CPU oracle (oracle/) to the reference: golden vectors emitted by the reference's own
code (tests/golden/make_golden.py) and, when present, the reference's compiled CPU entry points
(oracle/_ref).  CPU only."""
import os

import numpy as np
import pytest
import torch

from lidardetection_amd import synth
from oracle import c_oracle, pp_oracle, ref_loader


@pytest.fixture(scope="module")
def g_iou(golden_dir):
    return np.load(os.path.join(golden_dir, "iou3d_ref.npz"))


@pytest.fixture(scope="module")
def g_pp(golden_dir):
    return np.load(os.path.join(golden_dir, "pp_modules.npz"))


def test_iou_bev_oracle_bit_exact_vs_reference_golden(g_iou):
    out = c_oracle.pairwise(g_iou["boxes_a"], g_iou["boxes_b"], 1)
    ref = g_iou["iou_bev_cpu"]
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    assert (ref > 0).sum() > 200  # the fixture does exercise overlapping pairs


def test_iou_bev_oracle_on_nms_boxes(g_iou):
    b = g_iou["nms_boxes_sorted"]
    out = c_oracle.pairwise(b, b, 1)
    assert np.array_equal(out.view(np.uint32), g_iou["nms_iou_bev_cpu"].view(np.uint32))


def test_nms_mask_matches_thresholded_reference_iou(g_iou):
    """mask bit (i, j>i) == reference IoU(i,j) > thr, for the thresholds the configs use."""
    b = g_iou["nms_boxes_sorted"]
    ref = g_iou["nms_iou_bev_cpu"]
    n = len(b)
    for thr in (0.01, 0.1, 0.7):
        mask = c_oracle.nms_mask(b, thr)
        bits = np.unpackbits(mask.view(np.uint8), axis=1, bitorder="little")[:, :n].astype(bool)
        expect = np.triu(ref > np.float32(thr), k=1)
        assert np.array_equal(bits, expect)
        # greedy restatement vs. a direct python greedy on the reference matrix
        keep = c_oracle.nms_greedy(mask)
        alive = np.ones(n, bool)
        exp_keep = []
        for i in range(n):
            if alive[i]:
                exp_keep.append(i)
                alive[i + 1:] &= ~expect[i, i + 1:]
        assert keep.tolist() == exp_keep


@pytest.mark.skipif(ref_loader.load("iou3d_nms_cuda") is None, reason="oracle/_ref not built")
def test_iou_oracle_vs_live_reference_build():
    m = ref_loader.load("iou3d_nms_cuda")
    a = synth.boxes_random(101, 150)
    b = synth.boxes_random(102, 130)
    ref = torch.zeros(len(a), len(b))
    m.boxes_iou_bev_cpu(torch.from_numpy(a), torch.from_numpy(b), ref)
    out = c_oracle.pairwise(a, b, 1)
    assert np.array_equal(out.view(np.uint32), ref.numpy().view(np.uint32))


def test_pillar_vfe_oracle_vs_reference_module(g_pp):
    t = lambda k: torch.from_numpy(g_pp[k])
    out = pp_oracle.pillar_vfe(t("voxels"), t("num_points").float(), t("coords").float(), t("pfn_weight"),
                               t("bn_gamma"), t("bn_beta"), t("bn_mean"), t("bn_var"),
                               [float(x) for x in g_pp["voxel_size"]], [float(x) for x in g_pp["pc_range"]],
                               eps=float(g_pp["bn_eps"]))
    np.testing.assert_allclose(out.numpy(), g_pp["pillar_features"], rtol=0, atol=2e-6)


def test_mean_vfe_and_scatter_oracle_vs_reference_module(g_pp):
    mv = pp_oracle.mean_vfe(torch.from_numpy(g_pp["voxels"]), torch.from_numpy(g_pp["num_points"]).float())
    assert np.array_equal(mv.numpy(), g_pp["mean_features"])
    shp = tuple(g_pp["canvas_shape"])
    canvas = pp_oracle.pillar_scatter(torch.from_numpy(g_pp["pillar_features"]),
                                      torch.from_numpy(g_pp["coords"]).float(), shp[0], shp[3], shp[2])
    ref = np.zeros(shp, np.float32)
    ref[tuple(g_pp["canvas_nz_idx"])] = g_pp["canvas_nz_val"]
    assert np.array_equal(canvas.numpy(), ref)


def test_voxel_oracle_properties():
    """spconv is absent (parity unpinned): check the invariants of Appendix A.1 on the restatement."""
    pts = synth.cloud_ring(2000)
    rng, vs = synth.PP_RANGE, synth.PP_VOXEL
    vox, coords, num = c_oracle.voxelize(pts, vs, rng, 32, 16000)
    assert num.min() >= 1 and num.max() <= 32 and len(vox) == len(coords) == len(num)
    # first-appearance order + every stored point lies in its voxel + padded rows are zero
    lo, v = np.asarray(rng[:3], np.float32), np.asarray(vs, np.float32)
    cell = np.floor((pts[:, :3] - lo) / v).astype(np.int64)
    grid = np.round((np.asarray(rng[3:], np.float32) - lo) / v).astype(np.int64)
    ok = ((cell >= 0) & (cell < grid)).all(1)
    key = (cell[:, 2] * grid[1] + cell[:, 1]) * grid[0] + cell[:, 0]
    _, first = np.unique(key[ok], return_index=True)
    order = np.sort(first)
    exp_coords = cell[ok][order][:, ::-1]
    assert np.array_equal(coords, exp_coords[:16000].astype(np.int32))
    for vi in (0, 1, len(vox) // 2, len(vox) - 1):
        members = pts[ok][key[ok] == key[ok][order[vi]]][:32]
        assert np.array_equal(vox[vi, :len(members)], members)
        assert not vox[vi, len(members):].any() and num[vi] == len(members)
    # cap: uniform cloud has more pillars than max_voxels -> exactly max_voxels, later new voxels dropped
    pu = synth.cloud_uniform(1000)
    v2, c2, n2 = c_oracle.voxelize(pu, vs, rng, 32, 16000)
    assert len(v2) == 16000


def test_eval_iou_oracle_known_answers():
    """oracle/src/eval_iou_oracle.c (restated numba source, parity unpinned): closed-form cases."""
    from oracle import c_oracle
    a = np.array([[0, 0, 2, 4, 0]], np.float32)
    b = np.array([[1, 0, 2, 4, 0], [10, 10, 1, 1, 0.3]], np.float32)
    np.testing.assert_allclose(c_oracle.rotate_iou_eval(a, b, -1)[0], [4 / 12, 0.0], atol=1e-5)
    np.testing.assert_allclose(c_oracle.rotate_iou_eval(a, b, 2)[0], [4.0, 0.0], atol=1e-4)
    np.testing.assert_allclose(c_oracle.rotate_iou_eval(a, b[:1], 0)[0], [0.5], atol=1e-6)
    sq = np.array([[0, 0, 2, 2, 0]], np.float32)
    dia = np.array([[0, 0, 2, 2, np.pi / 4]], np.float32)            # square vs the same square turned by 45 deg: an octagon
    inter = 8 * (np.sqrt(2) - 1)
    np.testing.assert_allclose(c_oracle.rotate_iou_eval(sq, dia, 2)[0, 0], inter, atol=1e-4)
    np.testing.assert_allclose(c_oracle.rotate_iou_eval(sq, dia, -1)[0, 0], inter / (8 - inter), atol=1e-5)
    assert c_oracle.rotate_iou_eval(a[:0], b).shape == (0, 2)


@pytest.mark.parametrize("ksize,stride,padding,subm", [((3, 3, 3), (1, 1, 1), (1, 1, 1), True), ((3, 3, 3), (2, 2, 2), (1, 1, 1), False),
                                                       ((3, 3, 3), (2, 2, 2), (0, 1, 1), False), ((3, 1, 1), (2, 1, 1), (0, 0, 0), False)])
def test_sparse_oracle_equals_dense_conv_oracle(ksize, stride, padding, subm):
    """The full-grid oracle (oracle/spconv_sparse_oracle.py: binary search over the active sites + one fp64 matmul per kernel
    offset) against the dense conv3d oracle and the brute-force rulebook (oracle/spconv_oracle.py) on a grid small enough to
    densify: same output sites, same features, same number of pairs per offset."""
    from oracle import spconv_oracle as so, spconv_sparse_oracle as sp
    r = np.random.default_rng(11)
    shape, B, cin, cout = [9, 12, 10], 2, 5, 7
    cells = shape[0] * shape[1] * shape[2]
    pick = np.concatenate([np.sort(r.choice(cells, 160, replace=False)) + b * cells for b in range(B)])
    b_, rem = np.divmod(pick, cells)
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    idx = np.stack([b_, z, y, x], 1).astype(np.int64)
    idx = idx[r.permutation(len(idx))]                    # row order must not matter
    feats = r.standard_normal((len(idx), cin))
    w = r.standard_normal((*ksize, cin, cout))
    bias = r.standard_normal(cout)
    triples, outs = so.rulebook(idx, shape, list(ksize), list(stride), list(padding), subm)
    counts = np.bincount([t[0] for t in triples], minlength=int(np.prod(ksize)))
    if subm:
        got, n_k = sp.subm_conv(feats, idx, shape, w, bias, list(ksize))
        want = so.conv_features(feats, idx, B, shape, w, bias, list(ksize), [1, 1, 1], [0, 0, 0], True, idx).numpy()
    else:
        got, oidx, oshape, n_k = sp.sparse_conv(feats, idx, shape, w, bias, list(ksize), list(stride), list(padding))
        assert [tuple(int(v) for v in c) for c in oidx] == outs               # same active outputs, ascending (b, z, y, x)
        assert oshape == so.out_shape(shape, ksize, stride, padding)
        want = so.conv_features(feats, idx, B, shape, w, bias, list(ksize), list(stride), list(padding), False, np.array(outs)).numpy()
    assert n_k == counts.tolist()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


@pytest.mark.parametrize("ksize,stride,padding", [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (0, 1, 1)), ((3, 1, 1), (2, 1, 1), (0, 0, 0))])
def test_sparse_inverse_conv_oracle_equals_dense_transposed_conv_oracle(ksize, stride, padding):
    """spconv_sparse_oracle.inverse_conv (the paired convolution's pairs, swapped) against the dense conv_transpose3d oracle
    (oracle/spconv_oracle.py: inverse_conv_features) on a grid small enough to densify."""
    from oracle import spconv_oracle as so, spconv_sparse_oracle as sp
    r = np.random.default_rng(17)
    shape, B, cin, cout = [9, 12, 10], 2, 6, 5
    cells = shape[0] * shape[1] * shape[2]
    pick = np.concatenate([np.sort(r.choice(cells, 170, replace=False)) + b * cells for b in range(B)])
    b_, rem = np.divmod(pick, cells)
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    idx = np.stack([b_, z, y, x], 1).astype(np.int64)[r.permutation(len(pick))]
    _, oidx, oshape, _ = sp.sparse_conv(r.standard_normal((len(idx), 3)), idx, shape, r.standard_normal((*ksize, 3, 4)), None,
                                        list(ksize), list(stride), list(padding))
    oidx = oidx[r.permutation(len(oidx))]                  # the small tensor's rows in any order
    feats = r.standard_normal((len(oidx), cin))
    w = r.standard_normal((*ksize, cin, cout))
    bias = r.standard_normal(cout)
    got = sp.inverse_conv(feats, oidx, oshape, idx, shape, w, bias, list(ksize), list(stride), list(padding))
    want = so.inverse_conv_features(feats, oidx, B, oshape, w, bias, list(ksize), list(stride), list(padding), idx, shape).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


def test_spconv_nk_fixture_is_what_the_sparse_oracle_produces():
    """the sparse-conv FLOP count of the bench (`extra.spconv_gemm.gflop_useful`) is 2 * sum n_k * Cin * Cout with n_k from
    the ORACLE rulebook of the fixed synthetic batch, committed as tests/golden/spconv_nk_second_kitti_bs16.json.  Regenerates the table
    (C-oracle voxelisation + sparse fp64 oracle rulebooks, no GPU) and compares it entry by entry; SubM layers must be symmetric
    (n_k == n_{26-k}) and share their centre count with their row count."""
    import json
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    import make_nk_fixture
    want = json.load(open(os.path.join(here, "spconv_nk_second_kitti_bs16.json")))
    got = make_nk_fixture.build()
    assert got["voxels"] == want["voxels"]
    for a, b in zip(got["layers"], want["layers"]):
        assert a["layer"] == b["layer"] and a["rows_out"] == b["rows_out"] and a["n_k"] == b["n_k"], a["layer"]
        if a["kind"] == "subm":
            assert a["n_k"] == a["n_k"][::-1] and a["n_k"][13] == a["rows_out"]
    assert abs(got["gflop_useful_total"] - want["gflop_useful_total"]) < 1e-9
