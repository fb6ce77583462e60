"""How much does the ORDER of the mask groups matter for the mask-ordered GEMM?  Same tables, three row orders:
sorted by mask (torch.sort), equal masks grouped but groups in random order, and groups ordered by a 12-bit prefix then random."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth, spconv
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone, vfe
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.voxelizer import BatchVoxelizer
from lidardetection_amd.spconv import ops
dev = torch.device("cuda:0"); B = 16
frames = [synth.cloud_ring(2000 + f) for f in range(B)]
o = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000).voxelize_frames(frames, device=dev)
bd = {"voxels": o["voxels"], "voxel_num_points": o["voxel_num_points"], "voxel_coords": o["voxel_coords"], "batch_size": B}
bd = vfe.MeanVFE(AttrDict(), 4)(bd)
m = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()


def timeit(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def masks_of(nbr):
    K = nbr.shape[1]
    return ((nbr >= 0).long() << torch.arange(K, device=nbr.device)).sum(1)


with torch.no_grad():
    x = spconv.SparseConvTensor(bd["voxel_features"], bd["voxel_coords"].int(), m.sparse_shape, B)
    tot = [0.0, 0.0, 0.0, 0.0]
    g = torch.Generator(device="cpu").manual_seed(0)
    def walk(mod, x):
        for c in mod._modules.values():
            if isinstance(c, spconv.SparseSequential):
                x = walk(c, x)
            elif isinstance(c, spconv.SparseConvolution):
                y = c(x)
                nbr = y.indice_dict[c.indice_key]["nbr"]
                w = c.weight.reshape(-1, c.in_channels, c.out_channels).contiguous(); f = x.features.contiguous()
                if ops.sorted_gemm_supported(w.shape[0], c.in_channels, c.out_channels):
                    mk = masks_of(nbr)
                    mi = mk.int()
                    u, inv = torch.unique(mk, return_inverse=True)
                    o_sort = torch.argsort(mk).int()
                    rnd = torch.randperm(u.numel(), generator=g).to(dev)
                    o_rand = torch.argsort(rnd[inv]).int()
                    pre = (u >> 15)                                   # 12-bit prefix, then random inside
                    key2 = pre * (1 << 20) + rnd
                    o_pre = torch.argsort(key2[inv]).int()
                    ts = [timeit(lambda: ops.indice_conv_fused(f, nbr, w, None, None, True, (mi, oo))) for oo in (o_sort, o_rand, o_pre)]
                    t0 = timeit(lambda: ops.indice_conv_fused(f, nbr, w, None, None, True, None))
                    print(f"  {c.indice_key:13s} {c.in_channels:3d}->{c.out_channels:3d} rows {nbr.shape[0]:7d}: table {t0:7.1f} | sorted {ts[0]:7.1f} | grouped, random group order {ts[1]:7.1f} | 12-bit prefix + random {ts[2]:7.1f} us", flush=True)
                    for i, v in enumerate([t0] + ts): tot[i] += v
                x = y
            else:
                x.features = c(x.features)
        return x
    for name in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
        x = walk(getattr(m, name), x)
    print(f"totals: table {tot[0]/1e3:.3f} ms | sorted {tot[1]/1e3:.3f} | random groups {tot[2]/1e3:.3f} | prefix+random {tot[3]/1e3:.3f}")
