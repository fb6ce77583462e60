"""Generates the committed golden fixtures from the REFERENCE ITSELF.  Runs only in the build
container (needs /root/reference and oracle/_ref); the .npz outputs are data (inputs + expected
outputs) and are what travels to the GPU box.

  pp_modules.npz   <- the reference's own pure-torch modules, imported standalone on CPU:
                      PillarVFE (pcdet/models/backbones_3d/vfe/pillar_vfe.py), MeanVFE (mean_vfe.py),
                      PointPillarScatter (pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py)
  iou3d_ref.npz    <- the reference's own compiled CPU entry points (oracle/_ref, built by
                      oracle/build_ref.py from unmodified sources): boxes_iou_bev_cpu, points_in_boxes_cpu

  bev_head.npz     <- the reference's own BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py) forward on a small
                      configuration with random eval-mode BatchNorm statistics, and its own ResidualCoder.decode_torch
                      (pcdet/utils/box_coder_utils.py) on random anchors / encodings

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from lidardetection_amd import synth  # noqa: E402
from oracle import build_ref, c_oracle, ref_loader  # noqa: E402


def make_pp_modules():
    sys.path.insert(0, "/root/reference/pcdet/models/backbones_3d")
    sys.path.insert(0, "/root/reference/pcdet/models/backbones_2d")
    import vfe  # reference sub-package (bypasses pcdet/__init__.py)
    import map_to_bev

    pc_range = np.array([0.0, -3.2, -3.0, 7.68, 3.2, 1.0], np.float32)   # nx=48, ny=40, nz=1
    voxel_size = [0.16, 0.16, 4.0]
    P = 32
    frames = []
    for f in range(2):
        pts = synth.cloud_ring(seed=2000 + f)
        r = np.random.default_rng(77 + f)
        pts = pts[r.permutation(len(pts))[:3000]]
        pts[:, 0] *= 7.68 / 69.12           # squeeze into the small test range, keeps clustering
        pts[:, 1] *= 3.2 / 39.68
        frames.append(c_oracle.voxelize(pts, voxel_size, pc_range, P, 700))
    vox = np.concatenate([f[0] for f in frames], 0)
    num = np.concatenate([f[2] for f in frames], 0)
    coords = np.concatenate([np.pad(f[1], ((0, 0), (1, 0)), constant_values=i) for i, f in enumerate(frames)], 0)

    torch.manual_seed(0)
    cfg = types.SimpleNamespace(USE_NORM=True, WITH_DISTANCE=False, USE_ABSLOTE_XYZ=True, NUM_FILTERS=[64])
    m = vfe.PillarVFE(model_cfg=cfg, num_point_features=4, voxel_size=voxel_size, point_cloud_range=pc_range)
    bn = m.pfn_layers[0].norm
    with torch.no_grad():
        bn.running_mean.uniform_(-0.5, 0.5)
        bn.running_var.uniform_(0.5, 1.5)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    m.eval()
    bd = {"voxels": torch.from_numpy(vox), "voxel_num_points": torch.from_numpy(num).float(),
          "voxel_coords": torch.from_numpy(coords).float()}
    with torch.no_grad():
        bd = m(bd)
        pillar = bd["pillar_features"].clone()
        sc = map_to_bev.PointPillarScatter(types.SimpleNamespace(NUM_BEV_FEATURES=64), grid_size=(48, 40, 1))
        bd = sc(bd)
        canvas = bd["spatial_features"].clone()
        mv = vfe.MeanVFE(model_cfg=types.SimpleNamespace(), num_point_features=4)
        mean_feat = mv({"voxels": torch.from_numpy(vox), "voxel_num_points": torch.from_numpy(num).float()})["voxel_features"]
    # canvas is 97% zeros: store sparsely (nonzero cells) + its shape
    nz = canvas.numpy().nonzero()
    np.savez_compressed(
        os.path.join(HERE, "pp_modules.npz"),
        pc_range=pc_range, voxel_size=np.array(voxel_size, np.float32), max_points=P,
        voxels=vox, num_points=num, coords=coords,
        pfn_weight=m.pfn_layers[0].linear.weight.detach().numpy(),
        bn_gamma=bn.weight.detach().numpy(), bn_beta=bn.bias.detach().numpy(),
        bn_mean=bn.running_mean.numpy(), bn_var=bn.running_var.numpy(), bn_eps=np.float32(bn.eps),
        pillar_features=pillar.numpy(), mean_features=mean_feat.numpy(),
        canvas_shape=np.array(canvas.shape), canvas_nz_idx=np.stack(nz, 0).astype(np.int32),
        canvas_nz_val=canvas.numpy()[nz])
    print("pp_modules.npz", vox.shape, pillar.shape, canvas.shape)


def make_iou3d_ref():
    build_ref.build()
    iou = ref_loader.load("iou3d_nms_cuda")
    roi = ref_loader.load("roiaware_pool3d_cuda")
    r = np.random.default_rng(5)
    boxes_a = synth.boxes_random(11, 96, extent=12.0)
    boxes_b = synth.boxes_random(12, 80, extent=12.0)
    # hand-made edge cases: identical, touching, contained, 90-degree, tiny, far away
    special = np.array([
        [5, 5, 0, 4, 2, 1.5, 0.0], [5, 5, 0, 4, 2, 1.5, 0.0], [9, 5, 0, 4, 2, 1.5, 0.0],
        [5, 5, 0, 1, 0.5, 1.5, 0.3], [5, 5, 0, 4, 2, 1.5, np.pi / 2], [5, 5, 0, 2, 4, 1.5, 0.0],
        [5.01, 5, 0, 4, 2, 1.5, 1e-4], [100, 100, 0, 4, 2, 1.5, 1.0], [5, 5, 0, 1e-3, 1e-3, 1, 0.7],
        [5, 7, 0, 4, 2, 1.5, 0.0], [5, 6.99, 0, 4, 2, 1.5, np.pi], [3, 4, 0, 3.9, 1.6, 1.56, -2.5],
    ], np.float32)
    boxes_a = np.concatenate([special, boxes_a], 0)
    boxes_b = np.concatenate([special[::-1].copy(), boxes_b], 0)
    out = torch.zeros(len(boxes_a), len(boxes_b))
    iou.boxes_iou_bev_cpu(torch.from_numpy(boxes_a), torch.from_numpy(boxes_b), out)
    nb, sc = synth.boxes_nms(seed=3000, objects=48, copies=8)
    order = np.argsort(-sc, kind="stable")
    nb = nb[order]
    out_n = torch.zeros(len(nb), len(nb))
    iou.boxes_iou_bev_cpu(torch.from_numpy(nb), torch.from_numpy(nb), out_n)

    pts = r.uniform(-1, 13, (4000, 3)).astype(np.float32)
    pts[:, 2] = r.uniform(-2, 2, 4000)
    pib = torch.zeros(len(boxes_a), len(pts), dtype=torch.int32)
    roi.points_in_boxes_cpu(torch.from_numpy(boxes_a), torch.from_numpy(pts), pib)
    np.savez_compressed(os.path.join(HERE, "iou3d_ref.npz"), boxes_a=boxes_a, boxes_b=boxes_b,
                        iou_bev_cpu=out.numpy(), nms_boxes_sorted=nb, nms_iou_bev_cpu=out_n.numpy(),
                        pib_points=pts, pib_cpu=np.packbits(pib.numpy().astype(np.uint8), axis=1),
                        pib_shape=np.array(pib.shape))
    print("iou3d_ref.npz", out.shape, out_n.shape, int(pib.sum()))


def make_bev_head():
    import importlib.util

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    bev = load("_ref_base_bev_backbone", "/root/reference/pcdet/models/backbones_2d/base_bev_backbone.py")
    coder = load("_ref_box_coder_utils", "/root/reference/pcdet/utils/box_coder_utils.py")
    cfg = types.SimpleNamespace(LAYER_NUMS=[1, 2], LAYER_STRIDES=[2, 2], NUM_FILTERS=[16, 32], UPSAMPLE_STRIDES=[1, 2],
                                NUM_UPSAMPLE_FILTERS=[32, 32])
    cfg.get = lambda k, d=None: getattr(cfg, k, d)
    torch.manual_seed(3)
    m = bev.BaseBEVBackbone(cfg, input_channels=16)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(torch.empty(mod.num_features).uniform_(-0.3, 0.3, generator=g))
                mod.running_var.copy_(torch.empty(mod.num_features).uniform_(0.6, 1.4, generator=g))
                mod.weight.copy_(torch.empty(mod.num_features).uniform_(0.5, 1.5, generator=g))
                mod.bias.copy_(torch.empty(mod.num_features).uniform_(-0.3, 0.3, generator=g))
    m.eval()
    x = torch.randn(2, 16, 24, 20, generator=g)
    x[:, :, ::3] = 0
    with torch.no_grad():
        y = m({"spatial_features": x})["spatial_features_2d"]
    sd = {"bev." + k: v.numpy() for k, v in m.state_dict().items()}
    # ResidualCoder.decode_torch on KITTI-like anchors
    n = 500
    r = np.random.default_rng(9)
    anchors = np.concatenate([r.uniform(0, 70, (n, 1)), r.uniform(-40, 40, (n, 1)), r.uniform(-2, 0, (n, 1)),
                              r.uniform(0.5, 4.5, (n, 3)), r.choice([0.0, 1.57], (n, 1))], 1).astype(np.float32)
    enc = (r.standard_normal((n, 7)) * 0.3).astype(np.float32)
    dec = coder.ResidualCoder().decode_torch(torch.from_numpy(enc), torch.from_numpy(anchors))
    np.savez_compressed(os.path.join(HERE, "bev_head.npz"), bev_input=x.numpy(), bev_output=y.numpy(),
                        decode_anchors=anchors, decode_enc=enc, decode_out=dec.numpy(), **sd)
    print("bev_head.npz", tuple(y.shape), tuple(dec.shape), len(sd))


if __name__ == "__main__":
    make_pp_modules()
    make_iou3d_ref()
    make_bev_head()
