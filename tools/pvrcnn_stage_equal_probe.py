"""PV-RCNN (bs 8): where do the staged pass of tests/test_gpu_configs.py and the assembled forward stop being bit-identical?
Runs the stage methods by hand — with the SAME functions the forward uses, folded GEMM chains included — and the assembled
forward with every stage method wrapped to record what it returned; prints torch.equal per stage, in order."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.pvrcnn import PVRCNNKitti
dev = torch.device("cuda:0"); B = 8
frames = [synth.cloud_ring(2000 + f) for f in range(B)]
sizes = [len(f) for f in frames]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
torch.manual_seed(0)
m = PVRCNNKitti(batch_size=B, n_max=max(sizes), device=dev).randomize_for_bench(3)


def flat(x):
    if torch.is_tensor(x): return [x]
    if isinstance(x, dict): return [t for k in sorted(x) for t in flat(x[k])]
    if isinstance(x, (list, tuple)): return [t for e in x for t in flat(e)]
    if hasattr(x, "features") and hasattr(x, "indices"): return [x.features, x.indices]
    return []


with torch.no_grad():
    m(pts, offs, sizes)                                    # warm-up
    names = ["keypoints", "trunk", "proposals", "set_abstraction", "roi_head", "final_nms"]
    rec = {}
    orig = {n: getattr(m, n) for n in names}
    for n in names:
        def wrap(fn, n=n):
            def inner(*a, **k):
                out = fn(*a, **k)
                rec[n] = [t.clone() for t in flat(out)]
                return out
            return inner
        setattr(m, n, wrap(orig[n]))
    m._dense_rec = None
    out_asm = m(pts, offs, sizes)
    torch.cuda.synchronize()
    asm = dict(rec)
    for n in names: setattr(m, n, orig[n])
    # staged, by hand, in the test's order (trunk, proposals, keypoints, ...)
    st = {}
    multi_scale, bev, head = m.trunk(pts, offs); st["trunk"] = flat((multi_scale, bev, head))
    p = m.proposals(head); st["proposals"] = flat(p); rois, roi_scores, roi_labels = p[0], p[1], p[2]
    kp = m.keypoints(pts, offs, sizes); st["keypoints"] = flat(kp)
    before, fused = m.set_abstraction(pts, sizes, kp, multi_scale, bev); st["set_abstraction"] = flat((before, fused))
    for how in ("module sequence (what the test used)", "folded chain (what the forward uses)"):
        logits = m.point_cls_layers(before) if how.startswith("module") else m._dense["point_cls_layers"](before)
        point_scores = torch.sigmoid(logits).max(dim=-1)[0]
        rh = m.roi_head(rois, kp, fused, point_scores); st["roi_head"] = flat(rh)
        fn = m.final_nms(rh[0], rh[1], roi_labels); st["final_nms"] = flat(fn)
        print(f"point scores from the {how}:")
        for n in names:
            a, b = asm[n], st[n]
            eq = len(a) == len(b) and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(a, b))
            worst = max((float((x.float() - y.float()).abs().max()) for x, y in zip(a, b) if x.shape == y.shape and x.numel()), default=0.0)
            print(f"   {n:16s} bit-identical: {eq}   (max |diff| {worst:.3e})")
