"""lidar_topk_desc vs torch.topk on the headline step's own masked scores (16 x 321 408, k = 4096) and on a tie mass."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import anchor_post
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
B, n, k, thr = 16, 321408, 4096, 0.1
def ms(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
for name, s in (("sigmoid(N(0,0.2)) scores (the bench's range)", torch.sigmoid(torch.randn(B, n, generator=g) * 0.2)),
                ("one tie mass of 0.47 below 1 % distinct scores", torch.where(torch.rand(B, n, generator=g) < 0.01, torch.rand(B, n, generator=g) * 0.5 + 0.5, torch.tensor(0.47)))):
    s = s.to(dev).contiguous()
    s = torch.where(s >= thr, s, torch.full_like(s, -1.0))
    t_hip = ms(lambda: anchor_post.topk_desc(s, k, thr))
    t_torch = ms(lambda: torch.topk(s, k, dim=1))
    print(f"{name}: lidar_topk_desc (hist + collect + finalize) {t_hip:.0f} us, torch.topk {t_torch:.0f} us")
