"""Per-layer A/B on the SECOND-KITTI tables (bs 16, ring clouds): implicit GEMM in table order vs on mask-sorted rows."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth, spconv
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone, vfe
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.voxelizer import BatchVoxelizer
from lidardetection_amd.spconv import ops

dev = torch.device("cuda:0")
B = 16
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


def eff32(nbr):
    n, K = nbr.shape; pad = (-n) % 32
    v = torch.nn.functional.pad((nbr >= 0), (0, 0, 0, pad)).view(-1, 32, K).any(1).sum()
    return float((nbr >= 0).sum()) / float(v * 32)


ops.PACKED_GEMM[0] = True      # this tool measures both kernels
with torch.no_grad():
    x = spconv.SparseConvTensor(bd["voxel_features"], bd["voxel_coords"].int(), m.sparse_shape, B)
    tot = [0.0, 0.0, 0.0, 0.0, 0.0]
    def walk(mod, x):
        for c in mod._modules.values():
            if isinstance(c, spconv.SparseSequential):
                x = walk(c, x)
            elif isinstance(c, spconv.SparseConvolution):
                y = c(x)
                nbr = y.indice_dict[c.indice_key]["nbr"]
                nk = int((nbr >= 0).sum()); fl = 2.0 * nk * c.in_channels * c.out_channels
                w = c.weight.reshape(-1, c.in_channels, c.out_channels).contiguous(); f = x.features.contiguous()
                t_old = timeit(lambda: ops.indice_conv_fused(f, nbr, w, None, None, True, None))
                line = f"  {c.indice_key:13s} {c.in_channels:3d}->{c.out_channels:3d} rows {nbr.shape[0]:7d} pairs/row {nk / nbr.shape[0]:5.2f}  table order {t_old:7.1f} us {fl / t_old / 1e6:6.1f} TF (row eff {eff32(nbr):.2f})"
                tot[0] += t_old; tot[2] += fl
                if ops.sorted_gemm_supported(w.shape[0], c.in_channels, c.out_channels):
                    t_sort = timeit(lambda: ops.mask_order(nbr))
                    st = ops.mask_order(nbr)
                    t_new = timeit(lambda: ops.indice_conv_fused(f, nbr, w, None, None, True, st))
                    line += f" | mask-sorted {t_new:7.1f} us {fl / t_new / 1e6:6.1f} TF (row eff {eff32(nbr[st[1].long()]):.2f}; sort {t_sort:6.1f} us)"
                    tot[1] += t_new; tot[3] += t_sort
                    pk = ops.pack_gemm_weights(w)
                    t_pk = timeit(lambda: ops.indice_conv_fused(f, nbr, w, None, None, True, st, pk)) if pk is not None else t_new
                    line += f" | packed {t_pk:7.1f} us {fl / t_pk / 1e6:6.1f} TF"
                    tot[4] += t_pk
                else:
                    tot[1] += t_old; tot[4] += t_old
                print(line, flush=True)
                x = y
            else:
                x.features = c(x.features)
        return x
    for name in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
        x = walk(getattr(m, name), x)
    print(f"total table-order {tot[0] / 1e3:.3f} ms ({tot[2] / tot[0] / 1e6:.1f} TF)  mask-sorted {tot[1] / 1e3:.3f} ms "
          f"({tot[2] / tot[1] / 1e6:.1f} TF = {tot[2] / tot[1] / 1e6 / 157.3 * 100:.1f}% of fp32 MFMA peak); sorts (every table, incl. reused) {tot[3] / 1e3:.3f} ms")
    print(f"packed-weight kernel {tot[4] / 1e3:.3f} ms ({tot[2] / tot[4] / 1e6:.1f} TF = {tot[2] / tot[4] / 1e6 / 157.3 * 100:.1f}% of fp32 MFMA peak)")
