"""Where the time of the two voxelise launches goes (library built with -DVXL_STAMPS: tools/build_stamped.sh, selected through
LIDAR_HIP_SO): shader-clock stamps of bin role 0's phases + 100 MHz wall-clock start / end of EVERY workgroup of both launches.
usage: LIDAR_HIP_SO=lidardetection_amd/csrc/liblidar_hip_stamps.so python tools/vx_phase_probe.py [--flush] [--resident] [--cloud ring]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.voxelizer import BatchVoxelizer
dev = torch.device("cuda:0"); B = 16
ring = "ring" in sys.argv
frames = [synth.cloud_ring(2000 + f) if ring else synth.cloud_uniform(1000 + f) for f in range(B)]
sizes = [len(f) for f in frames]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
vz = BatchVoxelizer(synth.PP_VOXEL, synth.PP_RANGE, 32, 16000, algo=3)
out = vz.alloc_outputs(B, dev)
resident = "--resident" in sys.argv
n_max = max(sizes); G = -(-n_max // 1280); CAP = 3072
al = lambda x: (x + 255) // 256 * 256
err_off = al(B * n_max * 4) + al(B * G * CAP * 4) + al(B * G * CAP * 16)          # vx_carve: flagw, stgi, stg4, then the error page
names = ["start", "A0 my keys + lds init", "A1 (wave 0)", "A1 barrier (slowest wave)", "B table + C offsets", "D chains", "E words+staging", "end"]
junk = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev) if "--flush" in sys.argv else None
nbin = 8 * G * ((B + 7) // 8)
ntile = -(-n_max // 1024)
for it in range(6):
    if junk is not None:
        junk.fill_(float(it))          # cold caches: the state inside a detector step
    vz(pts, offs, n_max, out=out, resident=resident)
    torch.cuda.synchronize()
    ws = list(vz._ws.values())[0][0]
    page = ws[err_off: err_off + 65536].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    st = page[16:24]
    d = (st[1:8] - st[0:7]) & 0xFFFFFFFF
    print(f"iter {it}: " + "  ".join(f"{n} {int(c)}" for n, c in zip(names[1:], d)) + f"  | total {int((st[7]-st[0]) & 0xFFFFFFFF)} cycles")
    wall = page[64:64 + 2 * (1024 + B * ntile)].reshape(-1, 2)
    kb = wall[:nbin + 256]
    t0 = kb[:, 0].min()
    us = lambda x: (x - t0) / 100.0                                               # 100 MHz wall clock -> us
    b, fl, em = kb[:nbin], kb[nbin:nbin + 256], wall[1024:1024 + B * ntile]
    q = lambda a: f"min {a.min():.1f} med {np.median(a):.1f} max {a.max():.1f}"
    print(f"   bin roles  start {q(us(b[:, 0]))} | end {q(us(b[:, 1]))} | dur {q((b[:, 1] - b[:, 0]) / 100.0)}")
    print(f"   fill roles start {q(us(fl[:, 0]))} | end {q(us(fl[:, 1]))} | dur {q((fl[:, 1] - fl[:, 0]) / 100.0)}")
    print(f"   emit wgs   start {q(us(em[:, 0]))} | end {q(us(em[:, 1]))} | dur {q((em[:, 1] - em[:, 0]) / 100.0)}")
    if it == 5:
        es = page[32:36]
        en = ["start -> word arrived + prefetches issued", "first barrier", "rank + first-point stores issued"]
        print("   emit wg (last frame, tile 5): " + "  ".join(f"{n} {int((es[i + 1] - es[i]) & 0xFFFFFFFF)}" for i, n in enumerate(en)))
        d = ((em[:, 1] - em[:, 0]) / 100.0).reshape(B, ntile)
        print("   emit dur by tile (mean over frames): " + " ".join(f"{x:.1f}" for x in d.mean(0)))
        print("   emit dur by frame (mean over tiles): " + " ".join(f"{x:.1f}" for x in d.mean(1)))
        print("   emit start by frame (mean): " + " ".join(f"{x:.1f}" for x in us(em[:, 0]).reshape(B, ntile).mean(1)))
    if it == 5 and "--emit-detail" in sys.argv:
        e0 = em[:, 0].min()
        order = np.argsort(em[:, 0])
        for k in order[::8]:
            print(f"      emit wg {k:4d} (f {k // ntile:2d} tile {k % ntile:2d}) start {(em[k,0]-e0)/100:.2f} dur {(em[k,1]-em[k,0])/100:.2f}")
