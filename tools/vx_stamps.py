import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth, _lib
from lidardetection_amd.voxelizer import BatchVoxelizer
dev = torch.device("cuda:0")
frames = [synth.cloud_uniform(1000 + f) for f in range(16)]
sizes = [len(f) for f in frames]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
_lib.lib()
raw = ctypes.CDLL(_lib.SO_PATH)
raw.lidar_debug_stamp_ptr.restype = ctypes.c_void_p
raw.lidar_debug_stamp_ptr.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
vz = BatchVoxelizer(synth.PP_VOXEL, synth.PP_RANGE, 32, 16000, algo=1)
out = vz.alloc_outputs(16, dev)
raw.lidar_debug_set(9)
for _ in range(5):
    vz(pts, offs, max(sizes), out=out)
torch.cuda.synchronize()
ws, nb = vz._workspace(16, max(sizes), dev)
p = raw.lidar_debug_stamp_ptr(ws.data_ptr(), 16, max(sizes), 16000)
off = p - ws.data_ptr()
st = ws[off:off + 128 * 16 * 8].view(torch.int64).view(128, 16).cpu().numpy()
d = np.diff(st[:, :7], axis=1)
print("phase cycles (median over 128 WGs): init, B1, B2, C, D, E")
print(np.median(d, axis=0), "total", np.median(st[:, 6] - st[:, 0]))
print("span of kernel in cycles (max end - min start):", st[:, 6].max() - st[:, 0].min())
