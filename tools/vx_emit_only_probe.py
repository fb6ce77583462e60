"""-DVXL_STAMPS build: run the voxeliser normally a few times, then ONLY its second launch (LIDAR_VXL_DEBUG_EMIT_ONLY=1 is read at
every call) and print how the emit workgroups' start times spread."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidardetection_amd import synth
from lidardetection_amd.voxelizer import BatchVoxelizer
dev = torch.device("cuda:0"); B = 16
frames = [synth.cloud_uniform(1000 + f) for f in range(B)]
sizes = [len(f) for f in frames]
pts = torch.from_numpy(np.concatenate(frames)).to(dev)
offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
vz = BatchVoxelizer(synth.PP_VOXEL, synth.PP_RANGE, 32, 16000, algo=3)
out = vz.alloc_outputs(B, dev)
n_max = max(sizes); G = -(-n_max // 1280); CAP = 3072
al = lambda x: (x + 255) // 256 * 256
err_off = al(B * n_max * 4) + al(B * G * CAP * 4) + al(B * G * CAP * 16)
ntile = -(-n_max // 1024)
def stamps():
    torch.cuda.synchronize()
    ws = list(vz._ws.values())[0][0]
    page = ws[err_off: err_off + 65536].view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    em = page[64:64 + 2 * (1024 + B * ntile)].reshape(-1, 2)[1024:1024 + B * ntile]
    st = (em[:, 0] - em[:, 0].min()) / 100.0
    return f"late (> 2 us) {int((st > 2).sum())} of {len(st)}; start max {st.max():.2f}; dur med {np.median(em[:,1]-em[:,0])/100:.2f}; span {(em[:,1].max()-em[:,0].min())/100:.2f} us"
for _ in range(3):
    vz(pts, offs, n_max, out=out, resident=True)
print("both launches :", stamps())
os.environ["LIDAR_VXL_DEBUG_EMIT_ONLY"] = "1"
for _ in range(3):
    vz(pts, offs, n_max, out=out, resident=True)
    print("emit alone    :", stamps())
