"""shader-clock stamps of tk_finalize_kernel's phases (library built with -DTK_STAMPS: tools/build_stamped.sh)
usage: LIDAR_HIP_SO=lidardetection_amd/csrc/liblidar_hip_stamps.so python tools/topk_phase_probe.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import anchor_post
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
B, n, k, thr = 16, 321408, 4096, 0.1
s = torch.sigmoid(torch.randn(B, n, generator=g) * 0.2).to(dev)
al = lambda x: (x + 255) // 256 * 256
off_mm = al(B * 2048 * 4) + al(B * 64 * 4) + al(B * 4096 * 8) + al(B * 32 * 4)          # tk_carve: hist, nA, A, cB, then mm
names = ["histogram + counts in", "select bin", "list A in", "bin b1 in", "sort bin b1", "need best -> list", "sort the k winners", "write out"]
for it in range(4):
    anchor_post.topk_desc(s, k, thr)
    torch.cuda.synchronize()
    ws = anchor_post.topk_workspace(B, n, dev)
    st = ws[off_mm + 32 * 2 * 4: off_mm + 32 * 2 * 4 + 9 * 4].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    d = (st[1:] - st[:-1]) & 0xFFFFFFFF
    print("  ".join(f"{nm} {int(c)}" for nm, c in zip(names, d)) + f" | total {int((st[8] - st[0]) & 0xFFFFFFFF)} cycles")
