"""SECOND-KITTI sparse-conv stack micro-benchmark (BASELINE.json configs[2] shapes): rulebook + implicit-GEMM time per layer,
achieved fp32 TFLOP/s = 2 * sum_k n_k * Cin * Cout / time, vs the 157.3 TFLOP/s fp32 MFMA peak."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth, spconv
from lidardetection_amd.pcdet.models.backbones_3d import spconv_backbone, vfe
from lidardetection_amd.pcdet.utils.cfg import AttrDict
from lidardetection_amd.voxelizer import BatchVoxelizer
from lidardetection_amd.spconv import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
frames = [synth.cloud_ring(2000 + f) for f in range(B)]
vz = BatchVoxelizer(synth.SEC_VOXEL, synth.SEC_RANGE, 5, 16000)
o = vz.voxelize_frames(frames, device=dev)
bd = {"voxels": o["voxels"], "voxel_num_points": o["voxel_num_points"], "voxel_coords": o["voxel_coords"], "batch_size": B}
bd = vfe.MeanVFE(AttrDict(), 4)(bd)
m = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, [1408, 1600, 40]).to(dev).eval()
print(f"B={B} input voxels {bd['voxel_features'].shape[0]}")
with torch.no_grad():
    for _ in range(2):
        out = m(dict(bd))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        out = m(dict(bd))
    torch.cuda.synchronize()
    print(f"VoxelBackBone8x forward (rulebooks rebuilt each call): {(time.perf_counter()-t)/5*1e3:.2f} ms per batch of {B}")
    # per-layer implicit GEMM flops
    x = spconv.SparseConvTensor(bd["voxel_features"], bd["voxel_coords"].int(), m.sparse_shape, B)
    tot_flops, tot_t = 0.0, 0.0
    def walk(mod, x):
        global tot_flops, tot_t
        for c in mod._modules.values():
            if isinstance(c, spconv.SparseSequential):
                x = walk(c, x)
            elif isinstance(c, spconv.SparseConvolution):
                y = c(x)
                d = y.indice_dict[c.indice_key]
                nbr = d["nbr"]
                nk = int((nbr >= 0).sum())
                fl = 2.0 * nk * c.in_channels * c.out_channels
                w = c.weight.reshape(-1, c.in_channels, c.out_channels).contiguous()
                f = x.features.contiguous()
                for _ in range(3): ops._implicit_gemm(f, nbr, w, None, nbr.shape[0])
                torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): ops._implicit_gemm(f, nbr, w, None, nbr.shape[0])
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                tot_flops += fl; tot_t += ms
                def eff(T):   # useful fraction of MFMA rows if valid rows were packed per (T-row tile, offset)
                    n = nbr.shape[0]; pad = (-n) % T
                    v = torch.nn.functional.pad((nbr >= 0), (0, 0, 0, pad)).view(-1, T, nbr.shape[1]).sum(1)
                    return nk / float((((v + 31) // 32) * 32).sum())
                print(f"  {c.indice_key:13s} {c.in_channels:3d}->{c.out_channels:3d} rows {nbr.shape[0]:7d} pairs {nk:9d} "
                      f"({nk / nbr.shape[0]:5.2f}/row) {ms*1e3:8.1f} us {fl/ms/1e9:7.2f} TFLOP/s | MFMA row efficiency: "
                      f"now(32-row skip) {eff(32):.2f}, packed/128 {eff(128):.2f}, packed/256 {eff(256):.2f}")
                x = y
            else:
                x.features = c(x.features)
        return x
    for name in ("conv_input", "conv1", "conv2", "conv3", "conv4", "conv_out"):
        x = walk(getattr(m, name), x)
    print(f"implicit GEMM total: {tot_flops/1e9:.2f} GFLOP in {tot_t:.3f} ms = {tot_flops/tot_t/1e9:.2f} TFLOP/s ({tot_flops/tot_t/1e9/157.3*100:.1f}% of 157.3 TF fp32 MFMA peak)")
