import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth, _lib
from lidardetection_amd.ext import iou3d_nms_cuda
dev = torch.device("cuda:0")
b, s = synth.boxes_nms(seed=3000)
boxes = torch.from_numpy(b[np.argsort(-s, kind="stable")]).to(dev).view(1, 4096, 7)
_lib.lib(); raw = ctypes.CDLL(_lib.SO_PATH)
dbg = torch.zeros(64 * 64 * 8, dtype=torch.int64, device=dev)
raw.lidar_debug_nms_buffer.argtypes = [ctypes.c_void_p]
for _ in range(2):
    iou3d_nms_cuda.nms_batch(boxes, None, 0.01)
raw.lidar_debug_nms_buffer(dbg.data_ptr())
iou3d_nms_cuda.nms_batch(boxes, None, 0.01)
torch.cuda.synchronize()
raw.lidar_debug_nms_buffer(None)
st = dbg.view(64, 64, 8).cpu().numpy()
act = st[:, :, 0] > 0
d = np.diff(st[:, :, :6], axis=2)[act]
print("tiles", act.sum(), "phase cycles median: load, p1, p2, p3, store:", np.median(d, axis=0), "p90:", np.percentile(d, 90, axis=0))
tt = st[:, :, 6][act]
print("pairs after circle (median/max):", np.median(tt // 10000), (tt // 10000).max(), " after SAT:", np.median(tt % 10000), (tt % 10000).max())
print("kernel span cycles:", st[:, :, 5][act].max() - st[:, :, 0][act].min(), " sum of tile cycles:", (st[:, :, 5] - st[:, :, 0])[act].sum())
