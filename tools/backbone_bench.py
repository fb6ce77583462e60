"""Explores stock-torch options for the (out-of-scope) dense 2D backbone: memory format, BN folding."""
import os, sys, time, copy
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd.pointpillar import PointPillarKITTI
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
m = PointPillarKITTI(batch_size=16, device=dev).randomize_for_bench(0)
x = torch.randn(16, 64, 496, 432, device=dev)
x[:, :, ::3] = 0

def fold(seq):
    """Conv2d/ConvTranspose2d + BatchNorm2d (eval) -> conv with bias."""
    out, mods = [], list(seq)
    i = 0
    while i < len(mods):
        a = mods[i]
        if isinstance(a, (nn.Conv2d, nn.ConvTranspose2d)) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
            bn = mods[i + 1]
            s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            c = copy.deepcopy(a)
            if isinstance(a, nn.Conv2d):
                c.weight.data = a.weight.data * s.view(-1, 1, 1, 1)
            else:
                c.weight.data = a.weight.data * s.view(1, -1, 1, 1)
            c.bias = nn.Parameter(bn.bias.data - bn.running_mean * s)
            out.append(c); i += 2
        else:
            out.append(a); i += 1
    return nn.Sequential(*out)

def run(model, inp, n=10):
    with torch.no_grad():
        for _ in range(3): model.backbone_head(inp)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): model.backbone_head(inp)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

print("baseline NCHW            %.2f ms" % run(m, x))
m2 = copy.deepcopy(m)
m2.blocks = nn.ModuleList([fold(b) for b in m2.blocks]); m2.deblocks = nn.ModuleList([fold(b) for b in m2.deblocks])
with torch.no_grad():
    a = m.backbone_head(x); b = m2.backbone_head(x)
print("BN folded  max abs diff cls %.2e box %.2e" % ((a[0]-b[0]).abs().max().item(), (a[1]-b[1]).abs().max().item()))
print("BN folded NCHW           %.2f ms" % run(m2, x))
m3 = copy.deepcopy(m2).to(memory_format=torch.channels_last)
print("BN folded channels_last  %.2f ms" % run(m3, x.contiguous(memory_format=torch.channels_last)))
m4 = copy.deepcopy(m).to(memory_format=torch.channels_last)
print("baseline channels_last   %.2f ms" % run(m4, x.contiguous(memory_format=torch.channels_last)))
