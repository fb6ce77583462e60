"""GPU parity of the PointNet++ operators (stack + batch layouts) against the CPU oracle
(oracle/src/points_oracle.c).  Indices are bit-exact; gathered / interpolated features are bit-exact
where they are copies or single sequential expressions, 1e-5 where float atomics reorder a sum."""
import numpy as np
import pytest
import torch

from lidardetection_amd import synth
from lidardetection_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_modules as bmod
from lidardetection_amd.pcdet.ops.pointnet2.pointnet2_batch import pointnet2_utils as butils
from lidardetection_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_modules as smod
from lidardetection_amd.pcdet.ops.pointnet2.pointnet2_stack import pointnet2_utils as sutils
from oracle import c_oracle

pytestmark = pytest.mark.gpu


def _stack_scene(seed, sizes=(3000, 1777, 0, 2500), msizes=(400, 300, 0, 333)):
    r = np.random.default_rng(seed)
    xyz = np.concatenate([synth.cloud_ring(2000 + k)[:n, :3] for k, n in enumerate(sizes)], 0) if sum(sizes) else np.zeros((0, 3), np.float32)
    new = []
    off = 0
    for n, m in zip(sizes, msizes):
        if m:
            pick = r.choice(n, m, replace=False)
            new.append(xyz[off + pick] + r.normal(0, 0.05, (m, 3)).astype(np.float32))
        off += n
    new = np.concatenate(new, 0).astype(np.float32)
    return xyz.astype(np.float32), np.array(sizes, np.int32), new, np.array(msizes, np.int32)


@pytest.mark.parametrize("grid", [False, True])
@pytest.mark.parametrize("radius,nsample", [(0.4, 16), (1.6, 32), (0.01, 8)])
def test_ball_query_and_group_stack(dev, radius, nsample, grid, monkeypatch):
    """ball_query through both of its kernels (the public function picks by input size): exhaustive, and the cell grid"""
    from lidardetection_amd.ext import pointnet2_stack_cuda as native
    monkeypatch.setattr(native, "GRID_MIN_POINTS", 1 if grid else 1 << 30)
    xyz, xc, new, nc = _stack_scene(1)
    t = lambda a: torch.from_numpy(a).to(dev)
    idx_o = c_oracle.ball_query_stack(radius, nsample, xyz, xc, new, nc)
    idx, empty = sutils.ball_query(radius, nsample, t(xyz), t(xc), t(new), t(nc))
    exp_empty = idx_o[:, 0] == -1
    assert np.array_equal(empty.cpu().numpy(), exp_empty)
    idx_o[exp_empty] = 0
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    feat = np.random.default_rng(2).standard_normal((len(xyz), 37)).astype(np.float32)
    tf = t(feat).requires_grad_(True)
    g = sutils.grouping_operation(tf, t(xc), idx, t(nc))
    assert np.array_equal(g.detach().cpu().numpy(), c_oracle.group_points_stack(feat, xc, idx_o, nc))
    go = np.random.default_rng(3).standard_normal(g.shape).astype(np.float32)
    g.backward(t(go))
    # float atomics reorder the per-row sums (thousands of terms land on row 0 when every ball is empty)
    np.testing.assert_allclose(tf.grad.cpu().numpy(), c_oracle.group_points_grad_stack(go, idx_o, nc, xc, len(xyz)), rtol=1e-4, atol=2e-3)


def test_group_wide_tile_fallback(dev):
    xyz, xc, new, nc = _stack_scene(5, sizes=(500, 400), msizes=(50, 60))
    t = lambda a: torch.from_numpy(a).to(dev)
    idx, _ = sutils.ball_query(2.0, 64, t(xyz), t(xc), t(new), t(nc))
    feat = np.random.default_rng(6).standard_normal((len(xyz), 160)).astype(np.float32)   # 160 * 65 > LDS tile cap
    g = sutils.grouping_operation(t(feat), t(xc), idx, t(nc))
    assert np.array_equal(g.cpu().numpy(), c_oracle.group_points_stack(feat, xc, idx.cpu().numpy(), nc))


@pytest.mark.parametrize("n,m", [(20000, 2048), (5000, 512), (1000, 64), (700, 33), (40, 40), (25000, 128)])
def test_furthest_point_sampling(dev, n, m):
    r = np.random.default_rng(n)
    pts = np.stack([synth.cloud_uniform(1000 + k, n=n)[:, :3] for k in range(2)], 0)
    pts[1] = np.round(pts[1] * 2) / 2            # coarse lattice: many exact distance ties -> exercises the tie rule
    out = sutils.furthest_point_sample(torch.from_numpy(pts).to(dev), m)
    assert np.array_equal(out.cpu().numpy(), c_oracle.fps(pts, m))
    outb = butils.furthest_point_sample(torch.from_numpy(pts).to(dev), m)
    assert torch.equal(out, outb)


@pytest.mark.parametrize("n,m", [(19968, 2048), (4096, 64), (20480, 300), (7001, 2500)])
def test_furthest_point_sampling_bucketed_kernel(dev, n, m):
    """The bucketed kernel (4 096 <= n <= 20 480: spatial buckets, rounds skip every bucket the new sample cannot change) must
    give the reference's sample sequence bit for bit: KITTI-like ring clouds, a uniform cloud, a coarse lattice full of exact
    ties (the tie rule), a cloud with thousands of DUPLICATED points (zero distances, ties at 0), and the running distances
    left in `temp` must equal the oracle's as well."""
    ring = np.concatenate([synth.cloud_ring(2000)[:, :3], synth.cloud_ring(2001)[:, :3]], 0)[:n]
    if len(ring) < n:
        ring = np.concatenate([ring, synth.cloud_ring(2002)[:n - len(ring), :3]], 0)
    uni = synth.cloud_uniform(1000, n=n)[:, :3]
    lattice = np.round(synth.cloud_uniform(1001, n=n)[:, :3] * 2) / 2
    dup = synth.cloud_ring(2003)[:, :3][np.random.default_rng(5).integers(0, 3000, n)]
    pts = np.stack([ring, uni, lattice, dup], 0).astype(np.float32)
    out = sutils.furthest_point_sample(torch.from_numpy(pts).to(dev), m)
    want = c_oracle.fps(pts, m)
    for b, name in enumerate(("ring", "uniform", "lattice", "duplicates")):
        assert np.array_equal(out[b].cpu().numpy(), want[b]), name


def test_three_nn_and_interpolate_stack(dev):
    xyz, xc, new, nc = _stack_scene(7, sizes=(1200, 2, 900), msizes=(300, 2, 1))
    t = lambda a: torch.from_numpy(a).to(dev)
    d_o, i_o = c_oracle.three_nn_stack(new, nc, xyz, xc)
    d, i = sutils.three_nn(t(new), t(nc), t(xyz), t(xc))
    assert np.array_equal(i.cpu().numpy(), i_o)
    np.testing.assert_array_equal(d.cpu().numpy(), np.sqrt(d_o))       # batch with 2 known points: inf for the 3rd
    feat = np.random.default_rng(8).standard_normal((len(xyz), 19)).astype(np.float32)
    w = np.random.default_rng(9).uniform(0, 1, (len(new), 3)).astype(np.float32)
    tf = t(feat).requires_grad_(True)
    out = sutils.three_interpolate(tf, i, t(w))
    assert np.array_equal(out.detach().cpu().numpy(), c_oracle.three_interpolate_stack(feat, i_o, w))
    go = np.random.default_rng(10).standard_normal(out.shape).astype(np.float32)
    out.backward(t(go))
    np.testing.assert_allclose(tf.grad.cpu().numpy(), c_oracle.three_interpolate_grad_stack(go, i_o, w, len(xyz)), rtol=1e-5, atol=1e-5)


def test_batch_layout_ops(dev):
    r = np.random.default_rng(11)
    B, N, M, C = 3, 1500, 200, 21
    xyz = np.stack([synth.cloud_ring(2000 + k)[:N, :3] for k in range(B)], 0)
    new = xyz[:, r.choice(N, M, replace=False)] + r.normal(0, 0.05, (B, M, 3)).astype(np.float32)
    new = np.ascontiguousarray(new, np.float32)
    feat = r.standard_normal((B, C, N)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    idx = butils.ball_query(0.8, 16, t(xyz), t(new))
    idx_o = c_oracle.ball_query_batch(0.8, 16, xyz, new)
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    tf = t(feat).requires_grad_(True)
    g = butils.grouping_operation(tf, idx)
    assert np.array_equal(g.detach().cpu().numpy(), c_oracle.group_points_batch(feat, idx_o))
    go = r.standard_normal(g.shape).astype(np.float32)
    g.backward(t(go))
    np.testing.assert_allclose(tf.grad.cpu().numpy(), c_oracle.group_points_grad_batch(go, idx_o, N), rtol=1e-5, atol=1e-5)
    gi = r.integers(0, N, (B, M)).astype(np.int32)
    tf2 = t(feat).requires_grad_(True)
    ga = butils.gather_operation(tf2, t(gi))
    assert np.array_equal(ga.detach().cpu().numpy(), c_oracle.gather_points_batch(feat, gi))
    go2 = r.standard_normal(ga.shape).astype(np.float32)
    ga.backward(t(go2))
    np.testing.assert_allclose(tf2.grad.cpu().numpy(), c_oracle.gather_points_grad_batch(go2, gi, N), rtol=1e-5, atol=1e-5)
    d, i = butils.three_nn(t(new), t(xyz))
    d_o, i_o = c_oracle.three_nn_batch(new, xyz)
    assert np.array_equal(i.cpu().numpy(), i_o) and np.array_equal(d.cpu().numpy(), np.sqrt(d_o))
    w = r.uniform(0, 1, (B, M, 3)).astype(np.float32)
    tf3 = t(feat).requires_grad_(True)
    o = butils.three_interpolate(tf3, i, t(w))
    assert np.array_equal(o.detach().cpu().numpy(), c_oracle.three_interpolate_batch(feat, i_o, w))
    go3 = r.standard_normal(o.shape).astype(np.float32)
    o.backward(t(go3))
    np.testing.assert_allclose(tf3.grad.cpu().numpy(), c_oracle.three_interpolate_grad_batch(go3, i_o, w, N), rtol=1e-5, atol=1e-5)


def _perturb_bn(module, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.copy_(torch.empty(m.num_features).uniform_(-0.2, 0.2, generator=g))
                m.running_var.copy_(torch.empty(m.num_features).uniform_(0.5, 1.5, generator=g))
                m.weight.copy_(torch.empty(m.num_features).uniform_(0.8, 1.2, generator=g))
                m.bias.copy_(torch.empty(m.num_features).uniform_(-0.1, 0.1, generator=g))
    return module.eval()


def _close(got, want, what):
    want = np.asarray(want, np.float64)
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(np.asarray(got, np.float64) - want).max())
    assert err <= 1e-4 * scale, f"{what}: {err:.3e} vs scale {scale:.3e}"          # north_star: 1e-4 fp32 on features


def test_stack_set_abstraction_and_fp_modules_vs_oracle(dev):
    """QueryAndGroup / StackSAModuleMSG / StackPointnetFPModule (pointnet2_stack/pointnet2_utils.py:119-155,
    pointnet2_modules.py:58-92,111-137) on a ragged stacked batch with an EMPTY frame and empty balls: grouped tensors bit-exact,
    module outputs vs the fp64 replay (oracle ball query -> group -> concat -> the same MLP in float64), 1e-4."""
    from oracle import sa_oracle
    xyz, xc, new, nc = _stack_scene(12, sizes=(800, 600, 0, 450), msizes=(64, 48, 0, 40))
    new[5] += 50.0                                                   # a centre with nothing in reach: the empty-ball path
    t = lambda a: torch.from_numpy(a).to(dev)
    feat = np.random.default_rng(7).standard_normal((len(xyz), 8)).astype(np.float32)
    for use_xyz in (True, False):
        qg = sutils.QueryAndGroup(0.8, 16, use_xyz=use_xyz)
        g, idx = qg(t(xyz), t(xc), t(new), t(nc), t(feat))
        g_o, idx_o, empty_o = sa_oracle.query_and_group(0.8, 16, xyz, xc, new, nc, feat, use_xyz)
        assert empty_o[5] and np.array_equal(idx.cpu().numpy(), idx_o)
        assert np.array_equal(g.cpu().numpy(), g_o), f"QueryAndGroup(use_xyz={use_xyz})"
    g, _ = sutils.QueryAndGroup(0.8, 16)(t(xyz), t(xc), t(new), t(nc), None)          # xyz only
    assert np.array_equal(g.cpu().numpy(), sa_oracle.query_and_group(0.8, 16, xyz, xc, new, nc, None)[0])
    torch.manual_seed(0)
    sa = _perturb_bn(smod.StackSAModuleMSG(radii=[0.8, 1.6], nsamples=[16, 32], mlps=[[8, 16, 16], [8, 16, 32]]).to(dev), 1)
    with torch.no_grad():
        _, nf = sa(t(xyz), t(xc), t(new), t(nc), t(feat))
    assert nf.shape == (len(new), 48)
    _close(nf.cpu().numpy(), sa_oracle.stack_sa_msg(sa, xyz, xc, new, nc, feat), "StackSAModuleMSG")
    # no_grad + eval took the row-major inference path (forward_inference: lidar_group_rows_stack + GEMM chain); the
    # reference-layout path (autograd on) must agree with it and with the replay, with and without features / use_xyz
    with torch.no_grad():
        assert sa._inference_ready(t(xyz))
    assert not sa._inference_ready(t(xyz))
    _, nf_ref = sa(t(xyz), t(xc), t(new), t(nc), t(feat))
    _close(nf_ref.detach().cpu().numpy(), sa_oracle.stack_sa_msg(sa, xyz, xc, new, nc, feat), "StackSAModuleMSG, reference layout")
    sa_x = _perturb_bn(smod.StackSAModuleMSG(radii=[0.8], nsamples=[16], mlps=[[0, 16, 24]]).to(dev), 5)       # xyz only
    sa_f = _perturb_bn(smod.StackSAModuleMSG(radii=[1.6], nsamples=[8], mlps=[[8, 16]], use_xyz=False).to(dev), 6)
    with torch.no_grad():
        _close(sa_x(t(xyz), t(xc), t(new), t(nc), None)[1].cpu().numpy(), sa_oracle.stack_sa_msg(sa_x, xyz, xc, new, nc, None),
               "StackSAModuleMSG, xyz only")
        _close(sa_f(t(xyz), t(xc), t(new), t(nc), t(feat))[1].cpu().numpy(), sa_oracle.stack_sa_msg(sa_f, xyz, xc, new, nc, feat),
               "StackSAModuleMSG, use_xyz=False")
        # a first layer whose width is not a multiple of 4 takes the row-gather path (lidar_group_rows_stack) instead of the
        # layer-1-before-the-gather path (lidar_group_rows_affine_stack)
        sa_o = _perturb_bn(smod.StackSAModuleMSG(radii=[0.8], nsamples=[16], mlps=[[8, 6, 10]]).to(dev), 7)
        _close(sa_o(t(xyz), t(xc), t(new), t(nc), t(feat))[1].cpu().numpy(), sa_oracle.stack_sa_msg(sa_o, xyz, xc, new, nc, feat),
               "StackSAModuleMSG, odd first width")
        # 8 samples per ball = 4 queries per wave tile of the fused two-layer kernel, a query count that does not fill the
        # last tile, second-layer width 40 (two MFMA column tiles, the second partly empty)
        sa_q = _perturb_bn(smod.StackSAModuleMSG(radii=[1.0], nsamples=[8], mlps=[[8, 32, 40]]).to(dev), 8)
        nc_odd = nc.copy()
        nc_odd[-1] -= 1
        _close(sa_q(t(xyz), t(xc), t(new[:-1]), t(nc_odd), t(feat))[1].cpu().numpy(),
               sa_oracle.stack_sa_msg(sa_q, xyz, xc, new[:-1], nc_odd, feat), "StackSAModuleMSG, 8 samples, odd query count")
        # far from the origin the commuted first layer subtracts two large products: still inside the tolerance
        far = np.array([60.0, -35.0, 1.0], np.float32)
        _close(sa(t(xyz + far), t(xc), t(new + far), t(nc), t(feat))[1].cpu().numpy(),
               sa_oracle.stack_sa_msg(sa, xyz + far, xc, new + far, nc, feat), "StackSAModuleMSG, 70 m from the origin")
    fp = _perturb_bn(smod.StackPointnetFPModule(mlp=[48 + 8, 32]).to(dev), 2)
    with torch.no_grad():
        out = fp(t(xyz), t(xc), t(new), t(nc), unknown_feats=t(feat), known_feats=nf)
    assert out.shape == (len(xyz), 32)
    _close(out.cpu().numpy(), sa_oracle.stack_fp(fp, xyz, xc, new, nc, feat, nf.cpu().numpy()), "StackPointnetFPModule")
    # and gradients reach the features through the whole module (per-op backward parity is tested above)
    f = t(feat).requires_grad_(True)
    sa.train()
    _, nf2 = sa(t(xyz), t(xc), t(new), t(nc), f)
    nf2.sum().backward()
    assert f.grad is not None and torch.isfinite(f.grad).all() and float(f.grad.abs().sum()) > 0


def test_batch_set_abstraction_and_fp_modules_vs_oracle(dev):
    """PointnetSAModuleMSG / PointnetFPModule (pointnet2_batch/pointnet2_modules.py:61-101,124-170): FPS centres bit-exact,
    features vs the fp64 replay, 1e-4."""
    from oracle import sa_oracle
    B, N = 2, 1024
    bx = np.stack([synth.cloud_ring(2000 + k)[:N, :3] for k in range(B)], 0).astype(np.float32)
    bf = np.random.default_rng(8).standard_normal((B, 4, N)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    torch.manual_seed(1)
    bsa = _perturb_bn(bmod.PointnetSAModuleMSG(npoint=128, radii=[0.5, 1.0], nsamples=[8, 16], mlps=[[4, 8], [4, 16]]).to(dev), 3)
    with torch.no_grad():
        nx, nfeat = bsa(t(bx), t(bf))
    nx_o, nf_o = sa_oracle.batch_sa_msg(bsa, bx, bf)
    assert nx.shape == (B, 128, 3) and np.array_equal(nx.cpu().numpy(), nx_o)
    assert nfeat.shape == (B, 24, 128)
    _close(nfeat.cpu().numpy(), nf_o, "PointnetSAModuleMSG")
    bfp = _perturb_bn(bmod.PointnetFPModule(mlp=[24 + 4, 16]).to(dev), 4)
    with torch.no_grad():
        out = bfp(t(bx), nx, t(bf), nfeat)
    assert out.shape == (B, 16, N)
    _close(out.cpu().numpy(), sa_oracle.batch_fp(bfp, bx, nx_o, bf, nfeat.cpu().numpy()), "PointnetFPModule")


@pytest.mark.parametrize("ra,na,rb,nb", [(0.4, 16, 0.8, 16), (1.6, 32, 0.2, 8), (0.01, 4, 3.0, 64)])
def test_ball_query_two_radii_in_one_pass(dev, ra, na, rb, nb):
    """lidar_ball_query_stack2 (both scales of a StackSAModuleMSG in one pass over the distances) returns, for each radius,
    exactly the single-radius result — the oracle's (ball_query_gpu.cu:16-66): same indices, same -1 markers, ragged batch
    with an empty frame, a list that fills long before the other."""
    from lidardetection_amd.ext import pointnet2_stack_cuda as native
    xyz, xc, new, nc = _stack_scene(21)
    new[7] += 60.0                                                   # nothing in reach of either radius
    t = lambda a: torch.from_numpy(a).to(dev)
    ia = torch.zeros((len(new), na), dtype=torch.int32, device=dev)
    ib = torch.zeros((len(new), nb), dtype=torch.int32, device=dev)
    native.ball_query2_wrapper(len(xc), len(new), ra, na, rb, nb, t(new), t(nc), t(xyz), t(xc), ia, ib)
    for got, r, n in ((ia, ra, na), (ib, rb, nb)):
        want = c_oracle.ball_query_stack(r, n, xyz, xc, new, nc)
        got = got.cpu().numpy()
        empty = want[:, 0] == -1
        assert empty[7] and np.array_equal(got[:, 0] == -1, empty)
        assert np.array_equal(got[~empty], want[~empty])


@pytest.mark.parametrize("ra,na,rb,nb", [(0.4, 16, 0.8, 16), (1.6, 32, None, None), (0.05, 4, 6.0, 64), (2.4, 8, 1.2, 33)])
def test_ball_query_through_cell_grid(dev, ra, na, rb, nb):
    """lidar_ball_query_stack_grid (candidates binned into an x / y cell grid, three cell rows per centre, hits put back into
    index order by rank counting) returns exactly the exhaustive kernel's lists — the oracle's (ball_query_gpu.cu:16-66): ragged
    batch with an empty frame, centres far outside the candidates' bounding box, a NaN candidate, a ball with more hits than
    the 128-slot hit list holds (radius 6 m in the dense ring near the sensor -> the cut-down-and-tighten path, several times)."""
    from lidardetection_amd.ext import pointnet2_stack_cuda as native
    xyz, xc, new, nc = _stack_scene(33, sizes=(6000, 2500, 0, 4000), msizes=(500, 300, 0, 333))
    new[7] += 90.0                                                   # far outside the grid: nothing in reach
    new[11, :2] -= 300.0
    xyz = xyz.copy()
    xyz[17] = np.nan                                                 # never a hit, never a crash
    t = lambda a: torch.from_numpy(a).to(dev)
    ia = torch.zeros((len(new), na), dtype=torch.int32, device=dev)
    ib = torch.zeros((len(new), nb), dtype=torch.int32, device=dev) if rb else None
    native.ball_query_grid_wrapper(len(xc), len(new), ra, na, rb, nb, t(new), t(nc), t(xyz), t(xc), ia, ib)
    for got, r, n in ((ia, ra, na), (ib, rb, nb)):
        if got is None:
            continue
        want = c_oracle.ball_query_stack(r, n, xyz, xc, new, nc)
        got = got.cpu().numpy()
        empty = want[:, 0] == -1
        assert empty[7] and empty[11] and np.array_equal(got[:, 0] == -1, empty)
        assert np.array_equal(got[~empty], want[~empty]), (r, n)
    if rb == 6.0:                                                    # the case meant to overflow the 128-slot hit list does
        d2 = ((new[:500, None, :] - np.nan_to_num(xyz[None, :6000, :], nan=1e9)) ** 2).sum(-1)
        assert int((d2 < 36.0).sum(1).max()) > 512
