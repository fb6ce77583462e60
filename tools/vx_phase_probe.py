"""Phase stamps (shader clock) of bin role 0 of the fused voxelise launch (library built with -DVXL_STAMPS)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# needs a library built with -DVXL_STAMPS (HIPCC flags in csrc/build.py), selected through LIDAR_HIP_SO
from lidardetection_amd import synth
from lidardetection_amd.voxelizer import BatchVoxelizer
dev = torch.device("cuda:0"); B = 16
frames = [synth.cloud_uniform(1000 + f) for f in range(B)]
sizes = [len(f) for f in frames]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
vz = BatchVoxelizer(synth.PP_VOXEL, synth.PP_RANGE, 32, 16000, algo=3)
out = vz.alloc_outputs(B, dev)
n_max = max(sizes); G = -(-n_max // 2560)
al = lambda x: (x + 255) // 256 * 256
err_off = al(B * G * 6144 * 4) + al(B * n_max * 4) + al(B * n_max * 8)
names = ["start", "lds init", "A keys+append", "B2 table", "C offsets", "D chains", "E words+lists", "end"]
junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev) if "--flush" in sys.argv else None
for it in range(6):
    if junk is not None:
        junk.fill_(float(it))          # cold caches: the state inside a detector step
    vz(pts, offs, n_max, out=out)
    torch.cuda.synchronize()
    ws = list(vz._ws.values())[0][0]
    st = ws[err_off + 64: err_off + 64 + 32].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    d = (st[1:8] - st[0:7]) & 0xFFFFFFFF
    print(f"iter {it}: " + "  ".join(f"{n} {int(c)}" for n, c in zip(names[1:], d)) + f"  | total {int((st[7]-st[0]) & 0xFFFFFFFF)} ticks")
