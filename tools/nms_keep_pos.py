"""Where, in score order, the NMS_POST_MAXSIZE-th survivor sits for the bench's frames: sizes the first stage of the limited NMS
(csrc/iou3d.hip lidar_nms_batch_limited builds the suppression mask of the first 2 max_keep candidates only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lidardetection_amd.ext import iou3d_nms_cuda  # noqa: E402
from lidardetection_amd.pointpillar import PointPillarKITTI  # noqa: E402

dev = torch.device("cuda", 0)
frames, pts, offs, hoffs, n_max = bench.make_batch(16, 0, dev)
torch.manual_seed(0)
model = PointPillarKITTI(batch_size=16, max_voxels=16000, n_max=n_max, device=dev).randomize_for_bench(0)
seen = []
orig = iou3d_nms_cuda.nms_batch


def spy(boxes, counts, thresh, max_keep=None, **kw):
    keep, num = orig(boxes, counts, thresh, max_keep=max_keep, **kw)
    seen.append((counts.clone(), keep.clone(), num.clone(), max_keep))
    return keep, num


iou3d_nms_cuda.nms_batch = spy
with torch.no_grad():
    model(pts, offs, hoffs)
torch.cuda.synchronize()
counts, keep, num, mk = seen[-1]
last = [int(keep[f, int(num[f]) - 1]) if int(num[f]) > 0 else -1 for f in range(keep.shape[0])]
print("max_keep", mk, "counts", counts.tolist())
print("num_keep", num.tolist())
print("candidate index of the last survivor reported, per frame:", last, "max", max(last))
