"""GPU parity of the sparse-conv stack (lidardetection_amd.spconv) against oracle/spconv_oracle.py:
rulebooks as sets of (offset, in coord, out coord) triples + active output sets (bit-exact), features against
dense conv3d / conv_transpose3d within 1e-4 (north_star), gradients against autograd of the dense oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from lidardetection_amd import spconv
from oracle import spconv_oracle as so

pytestmark = pytest.mark.gpu


def _sites(seed, batch, shape, n):
    r = np.random.default_rng(seed)
    cells = batch * shape[0] * shape[1] * shape[2]
    pick = r.choice(cells, n, replace=False)
    b, rem = np.divmod(pick, shape[0] * shape[1] * shape[2])
    z, rem = np.divmod(rem, shape[1] * shape[2])
    y, x = np.divmod(rem, shape[2])
    # clustered occupancy (neighbours exist): fold coordinates towards a few blobs
    idx = np.stack([b, z, y, x], 1).astype(np.int32)
    return idx[r.permutation(n)]


def _table_triples(nbr, in_idx, out_idx):
    nbr, in_idx, out_idx = nbr.cpu().numpy(), in_idx.cpu().numpy(), out_idx.cpu().numpy()
    j, k = np.nonzero(nbr >= 0)
    return {(int(kk), tuple(int(v) for v in in_idx[nbr[jj, kk]]), tuple(int(v) for v in out_idx[jj])) for jj, kk in zip(j, k)}


CASES = [
    # (shape, ksize, stride, padding, subm, cin, cout, bias)
    ([9, 14, 16], [3, 3, 3], [1, 1, 1], [1, 1, 1], True, 4, 16, False),
    ([9, 14, 16], [3, 3, 3], [2, 2, 2], [1, 1, 1], False, 16, 32, False),
    ([5, 12, 10], [3, 3, 3], [2, 2, 2], [0, 1, 1], False, 64, 64, True),
    ([11, 8, 8], [3, 1, 1], [2, 1, 1], [0, 0, 0], False, 64, 128, False),
    ([6, 10, 10], [3, 3, 3], [1, 1, 1], [1, 1, 1], True, 5, 16, True),      # odd Cin (NuScenes' 5 point features)
    ([8, 9, 7], [3, 3, 3], [1, 1, 1], [0, 0, 0], True, 32, 32, False),
    # regular convolutions whose OUTPUT level has the INPUT level's shape: both would map to one pool grid
    ([9, 14, 16], [3, 3, 3], [1, 1, 1], [1, 1, 1], False, 16, 16, False),
    ([7, 8, 9], [3, 1, 1], [1, 1, 1], [1, 0, 0], False, 16, 32, True),
]


@pytest.mark.parametrize("case", CASES)
def test_rulebook_and_features(dev, case):
    shape, ks, st, pd, subm, cin, cout, bias = case
    B = 2
    n = int(0.35 * B * shape[0] * shape[1] * shape[2])
    idx = _sites(hash(tuple(shape)) % 1000, B, shape, n)
    feats = np.random.default_rng(1).standard_normal((n, cin)).astype(np.float32)
    torch.manual_seed(0)
    mod = (spconv.SubMConv3d if subm else spconv.SparseConv3d)(cin, cout, ks, stride=st, padding=pd, bias=bias, indice_key="k").to(dev)
    x = spconv.SparseConvTensor(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), shape, B)
    y = mod(x)
    d = x.indice_dict["k"]
    trip_o, outs_o = so.rulebook(idx, shape, ks, st, pd, subm)
    out_idx = y.indices.cpu().numpy()
    assert sorted(tuple(int(v) for v in r) for r in out_idx) == outs_o                 # active output set
    assert _table_triples(d["nbr"], x.indices, y.indices) == trip_o                     # forward table == brute force
    if not subm:   # transposed table describes the same relation
        nt = d["nbr_t"].cpu().numpy()
        i, k = np.nonzero(nt >= 0)
        tt = {(int(kk), tuple(int(v) for v in idx[ii]), tuple(int(v) for v in out_idx[nt[ii, kk]])) for ii, kk in zip(i, k)}
        assert tt == trip_o
        assert y.spatial_shape == so.out_shape(shape, ks, st, pd)
    ref = so.conv_features(feats, idx, B, shape, mod.weight.detach().cpu(), mod.bias.detach().cpu() if bias else None, ks, st, pd,
                           subm, out_idx)
    np.testing.assert_allclose(y.features.detach().cpu().double().numpy(), ref.numpy(), rtol=0, atol=1e-4)


def test_conv_backward_matches_dense_autograd(dev):
    shape, B, cin, cout = [7, 10, 9], 2, 16, 32
    n = 500
    idx = _sites(7, B, shape, n)
    feats = np.random.default_rng(2).standard_normal((n, cin)).astype(np.float32)
    for subm, ks, st, pd in ((True, [3, 3, 3], [1, 1, 1], [1, 1, 1]), (False, [3, 3, 3], [2, 2, 2], [1, 1, 1])):
        torch.manual_seed(1)
        mod = (spconv.SubMConv3d if subm else spconv.SparseConv3d)(cin, cout, ks, stride=st, padding=pd, bias=True).to(dev)
        f = torch.from_numpy(feats).to(dev).requires_grad_(True)
        y = mod(spconv.SparseConvTensor(f, torch.from_numpy(idx).to(dev), shape, B))
        go = torch.randn(y.features.shape, generator=torch.Generator().manual_seed(3)).to(dev)
        y.features.backward(go)
        # dense autograd oracle (float64)
        fd = torch.from_numpy(feats).double().requires_grad_(True)
        wd = mod.weight.detach().cpu().double().requires_grad_(True)
        bd = mod.bias.detach().cpu().double().requires_grad_(True)
        dense = torch.zeros((B, cin, *shape), dtype=torch.float64)
        ii = torch.from_numpy(idx).long()
        dense = dense.index_put((ii[:, 0], slice(None), ii[:, 1], ii[:, 2], ii[:, 3]), fd) if False else None
        dense = torch.zeros((B, *shape, cin), dtype=torch.float64).index_put((ii[:, 0], ii[:, 1], ii[:, 2], ii[:, 3]), fd).permute(0, 4, 1, 2, 3)
        o = F.conv3d(dense, wd.permute(4, 3, 0, 1, 2), bd, stride=(1 if subm else st), padding=([1, 1, 1] if subm else pd))
        oc = y.indices.cpu().long()
        (o[oc[:, 0], :, oc[:, 1], oc[:, 2], oc[:, 3]] * go.cpu().double()).sum().backward()
        np.testing.assert_allclose(f.grad.cpu().double().numpy(), fd.grad.numpy(), rtol=0, atol=1e-4)
        np.testing.assert_allclose(mod.weight.grad.cpu().double().numpy(), wd.grad.numpy(), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(mod.bias.grad.cpu().double().numpy(), bd.grad.numpy(), rtol=1e-4, atol=1e-3)


def test_inverse_conv_and_indice_key_reuse(dev):
    shape, B, c = [9, 12, 12], 2, 16
    n = 600
    idx = _sites(11, B, shape, n)
    feats = np.random.default_rng(4).standard_normal((n, c)).astype(np.float32)
    torch.manual_seed(2)
    down = spconv.SparseConv3d(c, 32, 3, stride=2, padding=1, bias=False, indice_key="spconv2").to(dev)
    sub = spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key="subm2").to(dev)
    sub_b = spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key="subm2").to(dev)
    up = spconv.SparseInverseConv3d(32, c, 3, indice_key="spconv2", bias=False).to(dev)
    x = spconv.SparseConvTensor(torch.from_numpy(feats).to(dev), torch.from_numpy(idx).to(dev), shape, B)
    y = down(x)
    y2 = sub_b(sub(y))
    assert y2.indice_dict["subm2"]["nbr"].data_ptr() == y.indice_dict["subm2"]["nbr"].data_ptr()   # rulebook reused, not rebuilt
    z = up(y2)
    assert torch.equal(z.indices, x.indices) and z.spatial_shape == shape
    ref = so.inverse_conv_features(y2.features.detach().cpu().numpy(), y2.indices.cpu().numpy(), B, y2.spatial_shape,
                                   up.weight.detach().cpu(), None, [3, 3, 3], [2, 2, 2], [1, 1, 1], idx, shape)
    np.testing.assert_allclose(z.features.detach().cpu().double().numpy(), ref.numpy(), rtol=0, atol=1e-4)


def test_dense_and_sequential_and_empty(dev):
    shape, B = [2, 20, 24], 2
    idx = _sites(13, B, shape, 300)
    f = torch.randn(300, 128, device=dev)
    x = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(dev), shape, B)
    dn = x.dense()
    ref = torch.zeros(B, 128, *shape)
    ii = torch.from_numpy(idx).long()
    ref[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]] = f.cpu()
    assert torch.equal(dn.cpu(), ref)
    x5 = spconv.SparseConvTensor(torch.randn(300, 5, device=dev), torch.from_numpy(idx).to(dev), shape, B)
    assert x5.dense().shape == (B, 5, *shape)                       # odd width -> torch path
    seq = spconv.SparseSequential(spconv.SubMConv3d(128, 16, 3, padding=1, bias=False, indice_key="a"),
                                  torch.nn.BatchNorm1d(16, eps=1e-3, momentum=0.01), torch.nn.ReLU()).to(dev)
    out = seq(x)
    assert out.features.shape == (300, 16) and float(out.features.min()) >= 0
    empty = spconv.SparseConvTensor(torch.zeros(0, 128, device=dev), torch.zeros(0, 4, dtype=torch.int32, device=dev), shape, B)
    oe = seq(empty)
    assert oe.features.shape[0] == 0
    dw = spconv.SparseConv3d(128, 16, 3, stride=2, padding=1).to(dev)(empty)
    assert dw.features.shape == (0, 16)


def _randomize_bn(m, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_(torch.empty(mod.num_features).uniform_(0.5, 1.5, generator=g))
                mod.bias.copy_(torch.empty(mod.num_features).uniform_(-0.3, 0.3, generator=g))
                mod.running_mean.copy_(torch.empty(mod.num_features).uniform_(-0.2, 0.2, generator=g))
                mod.running_var.copy_(torch.empty(mod.num_features).uniform_(0.6, 1.4, generator=g))


def test_inference_fused_conv_bn_relu_matches_module_sequence(dev):
    """no-grad path (BatchNorm folded into the weights, ReLU / residual in the GEMM epilogue, one launch per layer) vs the
    plain module sequence (conv kernel, torch BatchNorm1d, torch ReLU), for subm / strided / inverse / 1x1 / residual."""
    from lidardetection_amd.pcdet.models.backbones_3d.spconv_backbone import SparseBasicBlock, post_act_block
    from functools import partial
    shape, B, n = [9, 14, 12], 2, 700
    idx = _sites(21, B, shape, n)
    feats = torch.from_numpy(np.random.default_rng(5).standard_normal((n, 16)).astype(np.float32)).to(dev)
    norm = partial(torch.nn.BatchNorm1d, eps=1e-3, momentum=0.01)
    torch.manual_seed(4)
    net = spconv.SparseSequential(
        post_act_block(16, 16, 3, norm_fn=norm, padding=1, indice_key="subm1"),
        SparseBasicBlock(16, 16, norm_fn=norm, indice_key="res1"),
        post_act_block(16, 32, 3, norm_fn=norm, stride=2, padding=1, indice_key="spconv2", conv_type="spconv"),
        spconv.SparseSequential(spconv.SubMConv3d(32, 32, 1, bias=True), norm(32)),          # 1x1, BN, no ReLU
        post_act_block(32, 16, 3, norm_fn=norm, indice_key="spconv2", conv_type="inverseconv"),
        spconv.SubMConv3d(16, 8, 3, padding=1, bias=True, indice_key="subm1"),               # bare conv: unfused path
    ).to(dev).eval()
    _randomize_bn(net, 9)
    x = lambda: spconv.SparseConvTensor(feats, torch.from_numpy(idx).to(dev), shape, B)
    with torch.enable_grad():
        want = net(x())
    with torch.no_grad():
        got = net(x())
        got2 = net(x())                                   # folded weights come from the cache the second time
    assert torch.equal(got.indices, want.indices) and got.spatial_shape == want.spatial_shape
    scale = float(want.features.abs().max())
    assert float((got.features - want.features.detach()).abs().max()) <= 1e-5 * max(scale, 1.0)
    assert torch.equal(got.features, got2.features)
    # the cache notices a parameter update
    with torch.no_grad():
        net[0][1].bias.add_(0.5)
        moved = net(x())
    with torch.enable_grad():
        want2 = net(x())
    assert float((moved.features - want2.features.detach()).abs().max()) <= 1e-5 * max(scale, 1.0)
    assert not torch.equal(moved.features, got.features)


@pytest.mark.parametrize("cin,cout,n_out,p_valid", [(16, 16, 1000, 0.15), (16, 32, 257, 0.4), (32, 32, 128, 1.0),
                                                    (32, 64, 901, 0.35), (64, 64, 3000, 0.08), (64, 48, 130, 0.5),
                                                    (128, 128, 300, 0.3), (16, 64, 1, 1.0)])
def test_mask_sorted_gemm_is_bit_identical(dev, cin, cout, n_out, p_valid):
    """The fused GEMM on the mask-sorted table (skips padding MFMA tiles / unused offsets) must give exactly the bits of the
    table-order call (same products, same summation order), with bias / residual / ReLU; also checked against float64."""
    from lidardetection_amd.spconv import ops
    K, n_in = 27, 777
    g = torch.Generator(device="cpu").manual_seed(cin * 131 + cout + n_out)
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, K, generator=g) >= p_valid] = -1
    nbr[n_out // 2] = -1                                                  # a row without any neighbour
    feats = torch.randn(n_in, cin, generator=g)
    w = torch.randn(K, cin, cout, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    res = torch.randn(n_out, cout, generator=g)
    ref = torch.zeros(n_out, cout, dtype=torch.float64)
    for k in range(K):
        m = nbr[:, k] >= 0
        ref[m] += feats[nbr[m, k].long()].double() @ w[k].double()
    ref = torch.relu(ref + b.double() + res.double())
    nbr_d, f_d, w_d, b_d, r_d = (x.to(dev) for x in (nbr, feats, w, b, res))
    assert ops.sorted_gemm_supported(K, cin, cout)
    st = ops.mask_order(nbr_d)
    masks, perm = (x.cpu() for x in st)
    assert torch.equal(torch.sort(perm.long())[0], torch.arange(n_out))
    assert torch.equal(masks.long(), ((nbr >= 0).long() << torch.arange(K)).sum(1))
    ms = masks[perm.long()]
    assert int((ms[1:] != ms[:-1]).sum()) + 1 == torch.unique(masks).numel()      # every mask is one contiguous run of the order
    got = ops.indice_conv_fused(f_d, nbr_d, w_d, b_d, r_d, True, st)
    base = ops.indice_conv_fused(f_d, nbr_d, w_d, b_d, r_d, True, None)
    assert torch.equal(got, base)
    np.testing.assert_allclose(got.cpu().double().numpy(), ref.numpy(), rtol=0, atol=1e-4)
    # r04: the packed-weight kernel (LDS-DMA staging, 128-bit operand reads) sums the same products in the same order
    keep = ops.PACKED_GEMM[0]
    ops.PACKED_GEMM[0] = True                                             # (default off: measured slower; the kernel stays under test)
    try:
        packed = ops.pack_gemm_weights(w_d)
        assert packed is not None and packed.numel() == K * cin * ((cout + 31) // 32) * 32
        for rr, bb, relu in ((r_d, b_d, True), (None, None, False), (None, b_d, True)):
            want = ops.indice_conv_fused(f_d, nbr_d, w_d, bb, rr, relu, st)
            assert torch.equal(ops.indice_conv_fused(f_d, nbr_d, w_d, bb, rr, relu, st, packed), want)
        assert ops.pack_gemm_weights(torch.zeros(K, 4, 16, device=dev)) is None
    finally:
        ops.PACKED_GEMM[0] = keep
    assert not ops.sorted_gemm_supported(125, 16, 16) and not ops.sorted_gemm_supported(K, 4, 16)


@pytest.mark.parametrize("C,D", [(128, 2), (64, 1), (32, 3), (16, 4)])
def test_dense_bev_nhwc_equals_dense_view(dev, C, D):
    shape, B = [D, 20, 24], 2
    idx = _sites(31 + D, B, shape, 300)
    f = torch.randn(300, C, device=dev)
    x = spconv.SparseConvTensor(f, torch.from_numpy(idx).to(dev), shape, B)
    got = x.dense_bev()
    ref = torch.zeros(B, C, *shape)
    ii = torch.from_numpy(idx).long()
    ref[ii[:, 0], :, ii[:, 1], ii[:, 2], ii[:, 3]] = f.cpu()
    want = ref.view(B, C * D, shape[1], shape[2])
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("cin,cout,n_out,p_valid", [(16, 16, 1000, 0.15), (16, 32, 257, 0.4), (32, 64, 901, 0.35),
                                                    (64, 64, 5000, 0.1), (128, 128, 300, 0.3), (64, 16, 129, 1.0)])
def test_wgrad_mfma_matches_float64_and_the_scalar_kernel(dev, cin, cout, n_out, p_valid):
    """weight gradient on the matrix cores (with and without the mask order) vs a float64 gather-matmul and vs the
    VALU/atomics kernel it replaces; deterministic: two runs give the same bits."""
    from lidardetection_amd import _lib, workspace
    from lidardetection_amd.spconv import ops
    K, n_in = 27, 777
    g = torch.Generator(device="cpu").manual_seed(cin * 7 + cout + n_out)
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, K, generator=g) >= p_valid] = -1
    feats = torch.randn(n_in, cin, generator=g)
    go = torch.randn(n_out, cout, generator=g)
    ref = torch.zeros(K, cin, cout, dtype=torch.float64)
    for k in range(K):
        m = nbr[:, k] >= 0
        ref[k] = feats[nbr[m, k].long()].double().t() @ go[m].double()
    nbr_d, f_d, g_d = nbr.to(dev), feats.to(dev), go.to(dev)
    L = _lib.lib()
    assert L.lidar_spconv_wgrad_mfma_supported(K, cin, cout)
    wsb = L.lidar_spconv_wgrad_workspace_bytes(n_out, K, cin, cout)
    ws = workspace.get("spconv_wgrad_test", wsb, dev)
    outs = []
    for order in (None, ops.mask_order(nbr_d)[1], None):
        gw = torch.full((K, cin, cout), float("nan"), device=dev)
        _lib.check(L.lidar_spconv_wgrad_mfma(_lib.ptr(f_d), _lib.ptr(g_d), _lib.ptr(nbr_d), _lib.ptr(order), n_out, K, cin, cout,
                                             _lib.ptr(gw), _lib.ptr(ws), wsb, _lib.stream()), "lidar_spconv_wgrad_mfma")
        outs.append(gw)
        scale = float(ref.abs().max())
        np.testing.assert_allclose(gw.cpu().double().numpy(), ref.numpy(), rtol=0, atol=2e-5 * max(scale, 1.0))
    assert torch.equal(outs[0], outs[2])                                   # deterministic
    old = torch.zeros((K, cin, cout), device=dev)
    _lib.check(L.lidar_spconv_wgrad(_lib.ptr(f_d), _lib.ptr(g_d), _lib.ptr(nbr_d), n_out, K, cin, cout, _lib.ptr(old), _lib.stream()),
               "lidar_spconv_wgrad")
    np.testing.assert_allclose(outs[0].cpu().numpy(), old.cpu().numpy(), rtol=0, atol=2e-4 * max(float(ref.abs().max()), 1.0))


def test_mask_order_groups_equal_masks_bins_ascending(dev):
    """lidar_spconv_mask_group (hand-written: order-preserving hash of the distinct masks + scan + scatter, no sort): the masks are
    exact, the order is a permutation, every distinct mask forms exactly ONE contiguous run, and the runs come HEAVY first —
    ascending (K - popcount, top 7 mask bits): the workgroups with the most offsets are dealt first, similar masks stay
    together (the grouping the mask-ordered GEMM needs: tools/group_order_probe.py, tools/mask_pop_probe.py).  Random tables (up to 150 k DISTINCT masks: far
    more than real rulebooks hold), a table dominated by one mask (atomic contention), tiny K, repeated calls on one workspace."""
    from lidardetection_amd.spconv import ops
    g = torch.Generator().manual_seed(5)
    cases = [(1, 27, 0.35), (63, 27, 0.35), (5000, 27, 0.35), (150001, 27, 0.35), (300000, 8, 0.35), (4097, 31, 0.5), (200000, 27, 0.02),
             (70000, 3, 0.6)]
    for n, K, p in cases * 2:
        nbr = torch.where(torch.rand((n, K), generator=g) < p, torch.randint(0, max(n, 1), (n, K), generator=g), -1).int().to(dev)
        masks, order = ops.mask_order(nbr)
        want = torch.zeros(n, dtype=torch.int64)
        for k in range(K):
            want |= (nbr[:, k].cpu() >= 0).long() << k
        assert torch.equal(masks.cpu().long() & 0xFFFFFFFF, want)
        o = order.cpu().long()
        assert torch.equal(torch.sort(o).values, torch.arange(n))
        seq = want[o]
        runs = int((seq[1:] != seq[:-1]).sum()) + 1 if n else 0
        assert runs == torch.unique(want).numel(), (n, K, "a mask is split over several runs")
        pop = torch.zeros_like(seq)
        for k in range(K):
            pop += (seq >> k) & 1
        top = (((K - pop) & 31) << 7) | ((seq >> max(K - 7, 0)) & 127)
        assert bool((top[1:] >= top[:-1]).all()), (n, K)
    m0, o0 = ops.mask_order(torch.empty((0, 27), dtype=torch.int32, device=dev))
    assert m0.numel() == 0 and o0.numel() == 0


@pytest.mark.parametrize("cout,n_out,K,p_valid", [(16, 1000, 27, 0.15), (16, 129, 27, 1.0), (32, 4097, 27, 0.3), (48, 300, 27, 0.4),
                                                  (16, 1, 27, 1.0), (16, 500, 8, 0.5), (64, 260, 1, 1.0)])
def test_input_layer_gemm(dev, cout, n_out, K, p_valid):
    """Cin == 4 (the networks' input layer): the single-stage kernel (all K offsets gathered once) against float64, with and
    without bias / residual / ReLU, ragged row counts, K = 27 / 8 / 1 tables and a row without neighbours."""
    from lidardetection_amd.spconv import ops
    n_in = 555
    g = torch.Generator(device="cpu").manual_seed(cout * 7 + n_out + K)
    nbr = torch.randint(0, n_in, (n_out, K), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, K, generator=g) >= p_valid] = -1
    nbr[n_out // 2] = -1
    feats = torch.randn(n_in, 4, generator=g)
    w = torch.randn(K, 4, cout, generator=g) * 0.3
    b = torch.randn(cout, generator=g)
    res = torch.randn(n_out, cout, generator=g)
    ref = torch.zeros(n_out, cout, dtype=torch.float64)
    for k in range(K):
        m = nbr[:, k] >= 0
        ref[m] += feats[nbr[m, k].long()].double() @ w[k].double()
    nbr_d, f_d, w_d, b_d, r_d = (x.to(dev) for x in (nbr, feats, w, b, res))
    plain = ops.indice_conv_fused(f_d, nbr_d, w_d, None, None, False, None)
    np.testing.assert_allclose(plain.cpu().double().numpy(), ref.numpy(), rtol=0, atol=1e-4)
    full = ops.indice_conv_fused(f_d, nbr_d, w_d, b_d, r_d, True, None)
    np.testing.assert_allclose(full.cpu().double().numpy(), torch.relu(ref + b.double() + res.double()).numpy(), rtol=0, atol=1e-4)
    assert torch.equal(full, ops.indice_conv_fused(f_d, nbr_d, w_d, b_d, r_d, True, None))      # deterministic


def test_dense_grid_rulebooks_equal_hash_rulebooks(dev):
    """The dense-index-grid builder (csrc/rulebook_grid.hip through ops.GridPool) must produce the SAME tensors as the hash-table
    builder — output sites in the same first-touch order, identical nbr / nbr_t / SubM tables — for every geometry the reference
    uses (spconv_backbone.py:76-116), over several forwards through the same persistent grids: a coordinate BUFFER that is
    overwritten in place between forwards (what a voxeliser's output buffer does), shrinking and growing site counts, duplicate
    coordinates, and a budget too small for the grid (fallback to the hash builder)."""
    from lidardetection_amd.spconv import ops
    geoms = [([9, 14, 16], [3, 3, 3], [2, 2, 2], [1, 1, 1]), ([5, 12, 10], [3, 3, 3], [2, 2, 2], [0, 1, 1]),
             ([11, 8, 8], [3, 1, 1], [2, 1, 1], [0, 0, 0]), ([41, 60, 52], [3, 3, 3], [2, 2, 2], [1, 1, 1]),
             ([9, 14, 16], [3, 3, 3], [1, 1, 1], [1, 1, 1]), ([7, 8, 9], [3, 1, 1], [1, 1, 1], [1, 0, 0])]   # same-shape output level
    B = 3
    for shape, ks, st, pd in geoms:
        cells = B * shape[0] * shape[1] * shape[2]
        buf = torch.zeros((int(0.3 * cells), 4), dtype=torch.int32, device=dev)         # reused coordinate buffer
        for rnd, frac in enumerate((0.3, 0.05, 0.2)):
            n = int(frac * cells)
            idx = _sites(100 * rnd + shape[0], B, shape, n)
            if rnd == 2:
                idx[5] = idx[3]                                                          # a duplicated coordinate: lowest row wins
            buf[:n] = torch.from_numpy(idx).to(dev)
            coords = buf[:n]
            res = {}
            for mode in ("grid", "hash"):
                ops.GridPool.ENABLED = mode == "grid"
                try:
                    d = {}
                    sub = ops.subm_rulebook(coords, shape, [3, 3, 3], B, d)
                    out_idx, nbr, nbr_t = ops.conv_rulebook(coords, B, shape, ks, st, pd, d)
                    oshape = ops.get_conv_output_size(shape, ks, st, pd)
                    sub2 = ops.subm_rulebook(out_idx, oshape, [3, 3, 3], B, d)           # the next level, off the grid the conv left
                finally:
                    ops.GridPool.ENABLED = True
                res[mode] = (sub, out_idx, nbr, nbr_t, sub2)
            for a, b, what in zip(res["grid"], res["hash"], ("subm table", "output sites", "nbr", "nbr_t", "next-level subm table")):
                assert torch.equal(a, b), (shape, ks, rnd, what)
    # a grid over the budget is not used (and nothing breaks)
    keep = ops.GridPool.MAX_BYTES_PER_GRID
    ops.GridPool.MAX_BYTES_PER_GRID = 1024
    try:
        shape = [7, 9, 11]
        coords = torch.from_numpy(_sites(3, B, shape, 200)).to(dev)
        assert ops.GRIDS._entry(dev, B, shape) is None
        a = ops.subm_rulebook(coords, shape, [3, 3, 3], B, {})
    finally:
        ops.GridPool.MAX_BYTES_PER_GRID = keep
    assert torch.equal(a, ops.subm_rulebook(coords, shape, [3, 3, 3], B, {}))


def test_padding_rows_and_transposed_table(dev):
    """Capacity-sized coordinate tensors (csrc/rulebook_grid.hip): rows with batch index -1 appended to a tensor change nothing
    for the real rows (same SubM table, same output sites in the same order, same nbr / nbr_t) and have no neighbours themselves;
    lidar_spconv_transpose_table rebuilds nbr_t from nbr alone."""
    from lidardetection_amd import _lib
    from lidardetection_amd.spconv import ops
    B, shape, ks, st, pd = 2, [9, 14, 16], [3, 3, 3], [2, 2, 2], [1, 1, 1]
    n = 900
    coords = torch.from_numpy(_sites(5, B, shape, n)).to(dev)
    padded = torch.cat([coords, torch.full((137, 4), -1, dtype=torch.int32, device=dev)])
    d0, d1 = {}, {}
    sub0 = ops.subm_rulebook(coords, shape, [3, 3, 3], B, d0)
    out0, nbr0, nbrt0 = ops.conv_rulebook(coords, B, shape, ks, st, pd, d0)
    sub1 = ops.subm_rulebook(padded, shape, [3, 3, 3], B, d1)
    out1, nbr1, nbrt1 = ops.conv_rulebook(padded, B, shape, ks, st, pd, d1)
    assert torch.equal(sub1[:n], sub0) and bool((sub1[n:] == -1).all())
    assert torch.equal(out1, out0) and torch.equal(nbr1, nbr0)
    assert torch.equal(nbrt1[:n], nbrt0) and bool((nbrt1[n:] == -1).all())
    datas = {"nbr": nbr0, "nbr_t": None, "in_indices": coords}
    assert torch.equal(ops.ensure_table_t(datas), nbrt0)
    # pad_rows: rows beyond the device count become padding rows, the others stay
    buf = coords.clone()
    num = torch.tensor([700], dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().lidar_spconv_grid_pad_rows(_lib.ptr(buf), _lib.ptr(num), n, _lib.stream()), "pad")
    assert torch.equal(buf[:700], coords[:700]) and bool((buf[700:] == -1).all())


def test_speculative_capacity_forward_is_exact_and_recovers_from_overflow(dev):
    """run_stages_pipelined without host read-backs (capacity-sized tables + padding rows, counts checked once at the end):
    every stage output equals the exact path bit for bit; a capacity that turns out too small (hints shrunk by hand) is
    detected, the forward is replayed on the exact path, and the persistent grids are clean afterwards."""
    from lidardetection_amd import pillar_ops, spconv, synth
    from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone
    from lidardetection_amd.pcdet.utils.cfg import AttrDict
    from lidardetection_amd.spconv import ops
    from lidardetection_amd.voxelizer import BatchVoxelizer
    vox = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000)
    torch.manual_seed(3)
    m = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
    stages = [getattr(m, nm) for nm in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out")]

    def inputs(seed, nf):
        o = vox.voxelize_frames([synth.cloud_ring(seed + f) for f in range(nf)], device=dev)
        return pillar_ops.mean_vfe(o["voxels"], o["voxel_num_points"]), o["voxel_coords"].int()

    def run(feats, coords, nf, speculate):
        with torch.no_grad():
            x = spconv.SparseConvTensor(feats, coords, m.sparse_shape, nf)
            outs = spconv.run_stages_pipelined(stages, x, speculate=speculate)
        return outs, x.indice_dict

    def same(got, want):
        for a, b in zip(got[0], want[0]):
            assert torch.equal(a.indices, b.indices) and torch.equal(a.features, b.features)
        for key, d in want[1].items():
            if key.startswith("__"):
                continue
            assert torch.equal(got[1][key]["nbr"], d["nbr"]) and torch.equal(got[1][key]["out_indices"], d["out_indices"])

    used = []
    orig = ops._finish_speculative
    ops._finish_speculative = lambda *a: (used.append(1), orig(*a))[1]
    try:
        ops._CAP_HINTS.clear()
        fa, ca = inputs(4000, 3)
        want_a = run(fa, ca, 3, False)
        assert not used and ops._CAP_HINTS                    # the exact path leaves hints behind
        same(run(fa, ca, 3, True), want_a)
        assert len(used) == 4                                 # four strided convolutions, none waited for
        fb, cb = inputs(5000, 3)                              # other clouds through the same hints and grids
        want_b = run(fb, cb, 3, False)
        same(run(fb, cb, 3, True), want_b)
        for k in list(ops._CAP_HINTS):
            ops._CAP_HINTS[k] *= 0.25                         # capacities far too small: must be noticed and replayed
        same(run(fa, ca, 3, True), want_a)
        same(run(fb, cb, 3, True), want_b)                    # hints re-learnt, grids clean
        same(run(fa, ca, 3, True), want_a)
        # the transposed table an inference forward skipped is produced on demand and equals the builder's
        got = run(fa, ca, 3, True)
        full = {}
        spconv.prebuild_rulebooks(stages, ca.contiguous(), m.sparse_shape, 3, full)
        for key, d in full.items():
            if not key.startswith("__") and not d["subm"]:
                assert got[1][key]["nbr_t"] is None and torch.equal(got[1][key]["nbr"], d["nbr"])
                assert torch.equal(ops.ensure_table_t(dict(got[1][key])), d["nbr_t"])
    finally:
        ops._finish_speculative = orig


def test_graphed_forward_equals_eager_forward(dev):
    """spconv.GraphedStages: the whole VoxelBackBone8x inference forward captured as one hipGraph over capacity-padded inputs.
    Every stage output (coordinates, order, feature bits) equals the eager exact path for inputs of different sizes replayed
    through the same graph; a frozen capacity that is too small is noticed (exact-path answer, graph rebuilt); changed weights
    rebuild the graph; an eager forward in between (which leaves its rows in the shared grids) does not disturb the replay."""
    from lidardetection_amd import pillar_ops, spconv, synth
    from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone
    from lidardetection_amd.pcdet.utils.cfg import AttrDict
    from lidardetection_amd.voxelizer import BatchVoxelizer
    vox = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000)
    torch.manual_seed(5)
    m = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
    stages = [getattr(m, nm) for nm in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out")]
    B = 3

    def inputs(seed, keep=1.0):
        o = vox.voxelize_frames([synth.cloud_ring(seed + f)[:int(keep * 20000)] for f in range(B)], device=dev)
        return pillar_ops.mean_vfe(o["voxels"], o["voxel_num_points"]), o["voxel_coords"].int()

    def eager(f, c):
        with torch.no_grad():
            return spconv.run_stages_pipelined(stages, spconv.SparseConvTensor(f, c, m.sparse_shape, B), speculate=False)

    def same(got, want):
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a.spatial_shape == b.spatial_shape and torch.equal(a.indices, b.indices) and torch.equal(a.features, b.features)

    g = spconv.GraphedStages(stages, m.sparse_shape, B, 4, 3 * 16000, dev)
    cases = [inputs(6000), inputs(6100, 0.6), inputs(6200), inputs(6000)]
    for f, c in cases:
        same(g(f, c), eager(f, c))
    assert g.replays == len(cases) and g.fallbacks == 0
    f, c = cases[1]
    same(eager(f, c), eager(f, c))                                 # an eager forward leaves rows in the grids ...
    same(g(*cases[2]), eager(*cases[2]))                           # ... and the replay still starts from clean ones
    # capacities far too small for this input: detected, answered by the exact path, graph rebuilt with more room
    g2 = spconv.GraphedStages(stages, m.sparse_shape, B, 4, 3 * 16000, dev, headroom=1.0)
    small = inputs(6300, 0.25)
    same(g2(*small), eager(*small))                                # captured on a small cloud: small frozen capacities
    same(g2(*cases[0]), eager(*cases[0]))
    assert g2.fallbacks == 1
    same(g2(*cases[0]), eager(*cases[0]))                          # rebuilt
    assert g2.fallbacks == 1
    # new weights -> new graph (the captured launches hold the old folded weights' addresses)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.25)
    want = eager(*cases[2])
    assert not torch.equal(want[-1].features, g.outs[-1].features[:want[-1].features.shape[0]])
    same(g(*cases[2]), want)
    # and through the backbone module's opt-in switch
    m.graph_capacity = 3 * 16000
    with torch.no_grad():
        out = m({"voxel_features": cases[0][0], "voxel_coords": cases[0][1], "batch_size": B})
    same([out["encoded_spconv_tensor"]], [eager(*cases[0])[-1]])


def test_grid_rows_with_a_batch_index_outside_the_grid_are_padding_rows(dev):
    """a coordinate row whose batch index is >= the batch size the grids were allocated for (stale / too small
    batch_size) used to index past the grid.  It is now skipped like a padding row: no neighbours, reaches no output, and every
    other row's tables are what they are without it."""
    from lidardetection_amd.spconv import ops
    B, shape = 2, [9, 14, 16]
    idx = _sites(21, B, shape, 400)
    bad = np.array([[B + 3, 1, 2, 3], [B, 0, 0, 0]], np.int32)
    good = torch.from_numpy(idx).to(dev)
    both = torch.from_numpy(np.concatenate([idx, bad], 0)).to(dev)
    sub_g = ops.subm_rulebook(good, shape, [3, 3, 3], B, {})
    sub_b = ops.subm_rulebook(both, shape, [3, 3, 3], B, {})
    assert torch.equal(sub_b[:len(idx)], sub_g) and bool((sub_b[len(idx):] == -1).all())
    og, ng, tg = ops.conv_rulebook(good, B, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], {})
    ob, nb, tb = ops.conv_rulebook(both, B, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], {})
    assert torch.equal(ob, og) and torch.equal(nb, ng)
    assert torch.equal(tb[:len(idx)], tg) and bool((tb[len(idx):] == -1).all())


def test_grid_pool_budget_evicts_least_recently_used_grids(dev):
    """the pool used to keep one grid per (batch, shape) forever.  With a total budget, grids that hold no rows are
    evicted least-recently-used first, and rulebooks built afterwards are unchanged."""
    from lidardetection_amd.spconv import ops
    keep = ops.GridPool.MAX_TOTAL_BYTES
    ops.GRIDS.reset()
    shapes = [[8, 10 + k, 12] for k in range(6)]
    cells = lambda s: 2 * s[0] * s[1] * s[2] * 4
    ops.GridPool.MAX_TOTAL_BYTES = cells(shapes[-1]) * 3 + 64
    try:
        want = {}
        for s_ in shapes:
            c = torch.from_numpy(_sites(5, 2, s_, 150)).to(dev)
            want[tuple(s_)] = (c, ops.subm_rulebook(c, s_, [3, 3, 3], 2, {}))
            ops.GRIDS.wipe_all()                                    # (a finished forward leaves nothing behind)
            assert sum(e[0].numel() * 4 for e in ops.GRIDS.grids.values()) <= ops.GridPool.MAX_TOTAL_BYTES
        assert len(ops.GRIDS.grids) <= 3 and (str(dev), 2, *shapes[-1]) in ops.GRIDS.grids and (str(dev), 2, *shapes[0]) not in ops.GRIDS.grids
        for s_, (c, nbr) in want.items():                           # evicted levels are simply rebuilt
            assert torch.equal(ops.subm_rulebook(c, list(s_), [3, 3, 3], 2, {}), nbr)
            ops.GRIDS.wipe_all()
    finally:
        ops.GridPool.MAX_TOTAL_BYTES = keep
        ops.GRIDS.reset()


def test_grid_pool_budget_holds_when_every_grid_still_holds_rows(dev):
    """after a forward every level's grid still holds its rows (wipes are lazy), so "evict empty grids only" evicted
    nothing and the pool grew past its budget.  Two batch sizes through the same levels WITHOUT wipe_all in between: the pool stays
    within MAX_TOTAL_BYTES, a grid that cannot fit at all sends its level to the hash builder, and every table is unchanged."""
    from lidardetection_amd.spconv import ops
    keep = ops.GridPool.MAX_TOTAL_BYTES
    ops.GRIDS.reset()
    shape, big = [9, 14, 16], [41, 60, 52]
    nbytes = lambda b, s: b * s[0] * s[1] * s[2] * 4
    ops.GridPool.MAX_TOTAL_BYTES = nbytes(4, shape) + nbytes(4, [5, 7, 8]) + 64        # one bs-4 forward's two levels, nothing more
    try:
        ref = {}
        for B in (4, 2, 4, 1):
            c = torch.from_numpy(_sites(40 + B, B, shape, 60 * B)).to(dev)
            d = {}
            got = (ops.subm_rulebook(c, shape, [3, 3, 3], B, d), *ops.conv_rulebook(c, B, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], d))
            assert ops.GRIDS.holds_rows()                                              # nothing wiped: the r03 failure case
            assert sum(e[0].numel() * 4 for e in ops.GRIDS.grids.values()) <= ops.GridPool.MAX_TOTAL_BYTES, B
            ops.GridPool.ENABLED = False
            try:
                d = {}
                want = (ops.subm_rulebook(c, shape, [3, 3, 3], B, d), *ops.conv_rulebook(c, B, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], d))
            finally:
                ops.GridPool.ENABLED = True
            for a, b in zip(got, want):
                assert torch.equal(a, b), B
            ref[B] = got
        # a level whose grid alone exceeds the whole budget: no grid, hash builder, same table
        c = torch.from_numpy(_sites(77, 2, big, 500)).to(dev)
        assert ops.GRIDS._entry(dev, 2, big) is None
        a = ops.subm_rulebook(c, big, [3, 3, 3], 2, {})
        ops.GridPool.ENABLED = False
        try:
            assert torch.equal(a, ops.subm_rulebook(c, big, [3, 3, 3], 2, {}))
        finally:
            ops.GridPool.ENABLED = True
        assert sum(e[0].numel() * 4 for e in ops.GRIDS.grids.values()) <= ops.GridPool.MAX_TOTAL_BYTES
    finally:
        ops.GridPool.MAX_TOTAL_BYTES = keep
        ops.GRIDS.reset()


def test_two_row_tile_gemm_kernel_is_bit_identical(dev, tmp_path):
    """LIDAR_SPCONV_RT2=1 (csrc/sparse_conv.hip sc_implicit_gemm_rega2_kernel: 64 rows per wave, every B operand read feeds two MFMAs) —
    measured slower than the default kernel and therefore off, but kept under test: a child process (the switch is read once per
    process) runs the same mask-sorted 64 -> 64 and 32 -> 64 layers and must produce the default kernel's bits"""
    import os
    import subprocess
    import sys
    code = r"""
import sys, torch
from lidardetection_amd.spconv import ops
dev = torch.device("cuda:0")
outs = []
for cin, cout, n_out in ((64, 64, 5000), (32, 64, 4099), (64, 48, 2300)):
    g = torch.Generator(device="cpu").manual_seed(cin + cout + n_out)
    nbr = torch.randint(0, 777, (n_out, 27), generator=g, dtype=torch.int32)
    nbr[torch.rand(n_out, 27, generator=g) >= 0.3] = -1
    f = torch.randn(777, cin, generator=g).to(dev); w = (torch.randn(27, cin, cout, generator=g) * 0.2).to(dev)
    b = torch.randn(cout, generator=g).to(dev); r = torch.randn(n_out, cout, generator=g).to(dev)
    nd = nbr.to(dev)
    st = ops.mask_order(nd)
    outs.append(ops.indice_conv_fused(f, nd, w, b, r, True, st).cpu())
    outs.append(ops.indice_conv_fused(f, nd, w, None, None, False, None).cpu())
torch.save(outs, sys.argv[1])
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for flag in ("0", "1"):
        out = str(tmp_path / f"rt2_{flag}.pt")
        env = dict(os.environ, LIDAR_SPCONV_RT2=flag, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=300)
        res.append(torch.load(out, weights_only=True))
    assert len(res[0]) == 6
    for a, b in zip(*res):
        assert torch.equal(a, b)
