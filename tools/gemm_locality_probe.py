"""Is the mask-ordered implicit GEMM bound by the latency of its row gathers?  Times each SECOND-KITTI layer as is and with
every gathered row index folded into the first 1024 rows (same masks, same flops, all gathers L2 hits)."""
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


with torch.no_grad():
    x = spconv.SparseConvTensor(bd["voxel_features"], bd["voxel_coords"].int(), m.sparse_shape, B)
    def walk(mod, x):
        for c in mod._modules.values():
            if isinstance(c, spconv.SparseSequential):
                x = walk(c, x)
            elif isinstance(c, spconv.SparseConvolution):
                y = c(x)
                nbr = y.indice_dict[c.indice_key]["nbr"]
                w = c.weight.reshape(-1, c.in_channels, c.out_channels).contiguous(); f = x.features.contiguous()
                if ops.sorted_gemm_supported(w.shape[0], c.in_channels, c.out_channels):
                    st = ops.mask_order(nbr)
                    t0 = timeit(lambda: ops.indice_conv_fused(f, nbr, w, None, None, True, st))
                    hot = torch.where(nbr >= 0, nbr % 1024, nbr)
                    t1 = timeit(lambda: ops.indice_conv_fused(f, hot, w, None, None, True, st))
                    fl = 2.0 * float((nbr >= 0).sum()) * c.in_channels * c.out_channels
                    print(f"  {c.indice_key:13s} {c.in_channels:3d}->{c.out_channels:3d} rows {nbr.shape[0]:7d}  real gathers {t0:7.1f} us "
                          f"{fl / t0 / 1e6:5.1f} TF | gathers folded into 1024 rows {t1:7.1f} us {fl / t1 / 1e6:5.1f} TF", flush=True)
                x = y
            else:
                x.features = c(x.features)
        return x
    for name in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
        x = walk(getattr(m, name), x)
