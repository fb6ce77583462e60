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
vz = BatchVoxelizer(synth.PP_VOXEL, synth.PP_RANGE, 32, 16000, algo=1)
out = vz.alloc_outputs(16, dev)
for dbg in [0, 1, 2, 3, 4, 0]:
    raw.lidar_debug_set(dbg)
    for _ in range(3):
        vz(pts, offs, max(sizes), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        vz(pts, offs, max(sizes), out=out)
    e1.record(); torch.cuda.synchronize()
    print(f"dbg {dbg}: {e0.elapsed_time(e1)/50*1e3:.1f} us (whole sequence; bin kernel exits after phase {dbg})")
