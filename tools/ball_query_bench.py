import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lidardetection_amd import synth
from lidardetection_amd.ext import pointnet2_stack_cuda as native
dev = torch.device("cuda:0")
def ev(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B = 8
frames = [synth.cloud_ring(2000 + f)[:, :3] for f in range(B)]
xyz = torch.from_numpy(np.concatenate(frames, 0)).to(dev).contiguous()
xc = torch.tensor([len(f) for f in frames], dtype=torch.int32, device=dev)
kp = torch.from_numpy(np.concatenate([f[np.random.default_rng(i).choice(len(f), 2048, replace=False)] for i, f in enumerate(frames)], 0)).to(dev).contiguous()
kc = torch.full((B,), 2048, dtype=torch.int32, device=dev)
for name, (q, qc, c, cc) in {"SA: 16384 keypoints vs 8 x 19968 raw points": (kp, kc, xyz, xc),
                             "RoI: 172800 grid pts vs 8 x 2048 keypoints": ((kp.repeat(11, 1)[:172800].view(-1, 3) + torch.randn(172800, 3, device=dev) * 0.7).view(8, -1, 3).contiguous().view(-1, 3), torch.full((B,), 21600, dtype=torch.int32, device=dev), kp, kc)}.items():
    M = q.shape[0]
    for ra, rb in ((0.4, 0.8), (0.8, 1.6)):
        ia = torch.zeros((M, 16), dtype=torch.int32, device=dev); ib = torch.zeros((M, 16), dtype=torch.int32, device=dev)
        t2 = ev(lambda: native.ball_query2_wrapper(B, M, ra, 16, rb, 16, q, qc, c, cc, ia, ib))
        a0, b0 = ia.clone(), ib.clone()
        tg = ev(lambda: native.ball_query_grid_wrapper(B, M, ra, 16, rb, 16, q, qc, c, cc, ia, ib))
        same = bool(torch.equal(a0, ia) and torch.equal(b0, ib))
        print(f"{name} r=({ra},{rb}): exhaustive two-radius {t2:.1f} us | cell grid {tg:.1f} us | equal {same}", flush=True)
