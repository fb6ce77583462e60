"""SECOND-KITTI forward + NMS on one MI355X (BASELINE.json configs[2]): batched HIP voxelise -> MeanVFE -> sparse 3D
backbone (rulebooks + mask-ordered MFMA implicit GEMM, fused BN/ReLU epilogues) -> .dense() / HeightCompression -> folded
BEV backbone + merged heads -> HIP anchor post-processing -> batched rotated NMS.

Topology from tools/cfgs/kitti_models/second.yaml: VoxelBackBone8x (pcdet/models/backbones_3d/spconv_backbone.py:68-163),
HeightCompression (pcdet/models/backbones_2d/map_to_bev/height_compression.py:10-26), BaseBEVBackbone with LAYER_NUMS [5, 5],
LAYER_STRIDES [1, 2], NUM_FILTERS [128, 256], UPSAMPLE_STRIDES [1, 2], NUM_UPSAMPLE_FILTERS [256, 256]
(base_bev_backbone.py:6-112), AnchorHeadSingle with the three KITTI anchor classes x 2 rotations.  Random-init weights;
everything downstream of the heads is shared with pointpillar.PointPillarKITTI."""
import torch
import torch.nn as nn

from . import pillar_ops, synth
from .bev_backbone import FoldedBEVBackbone
from .pcdet.models.backbones_3d import spconv_backbone
from .pcdet.utils.cfg import AttrDict
from .pointpillar import PointPillarKITTI, generate_anchors, make_bev_backbone
from .voxelizer import BatchVoxelizer, grid_size_of


class SECONDKitti(PointPillarKITTI):
    def __init__(self, batch_size=16, max_voxels=16000, n_max=20000, device="cuda", score_thresh=0.1, nms_thresh=0.01,
                 nms_pre=4096, nms_post=500):
        nn.Module.__init__(self)
        self.B, self.n_max = batch_size, n_max
        self.pc_range, self.voxel_size = synth.SEC_RANGE, synth.SEC_VOXEL
        self.grid = [int(v) for v in grid_size_of(self.voxel_size, self.pc_range)]          # [1408, 1600, 40]
        self.voxelizer = BatchVoxelizer(self.voxel_size, self.pc_range, 5, max_voxels, 4)
        self.backbone3d = spconv_backbone.VoxelBackBone8x(AttrDict(), 4, self.grid)
        self.blocks, self.deblocks = make_bev_backbone(cin=256, layer_nums=(5, 5), strides=(1, 2), filters=(128, 256),
                                                       up_strides=(1, 2), up_filters=(256, 256))
        self.num_class, self.num_anchor_per_loc, self.num_dir_bins = 3, 6, 2
        self.conv_cls = nn.Conv2d(512, self.num_anchor_per_loc * self.num_class, 1)
        self.conv_box = nn.Conv2d(512, self.num_anchor_per_loc * 7, 1)
        self.conv_dir_cls = nn.Conv2d(512, self.num_anchor_per_loc * self.num_dir_bins, 1)
        self.dir_offset, self.dir_limit_offset = 0.78539, 0.0
        self.score_thresh, self.nms_thresh, self.nms_pre, self.nms_post = score_thresh, nms_thresh, nms_pre, nms_post
        self.channels_last = self.fold_bn = True
        self.to(device).eval()
        for mod in (self.blocks, self.deblocks, self.conv_cls, self.conv_box, self.conv_dir_cls):
            mod.to(memory_format=torch.channels_last)       # 2D part only (the sparse weights are 5-D)
        self.anchors = generate_anchors(self.pc_range, (self.grid[1] // 8, self.grid[0] // 8), device)
        self._vox_out = self.voxelizer.alloc_outputs(batch_size, device)
        self._folded = self._bev = None

    # ---- stages --------------------------------------------------------------------------------
    def voxelize_vfe(self, points, point_offsets, host_offsets=None):
        vox = self.voxelizer(points, point_offsets, self.n_max, compact=True, out=self._vox_out, resident=self.resident_voxels,
                             host_offsets=host_offsets)
        total = int(vox["voxel_offsets"][self.B])               # the sparse stack needs exact row counts (one read-back)
        feats = pillar_ops.mean_vfe(vox["voxels"][:total], vox["voxel_num_points"][:total])
        return feats, vox["voxel_coords"][:total]

    def sparse_backbone(self, feats, coords):
        bd = self.backbone3d({"voxel_features": feats, "voxel_coords": coords, "batch_size": self.B})
        return bd["encoded_spconv_tensor"].dense_bev()          # (B, 128 * 2, 200, 176), channels-last, one pass

    def backbone_head(self, canvas):
        return (self._bev_folded().merged(canvas),)

    @torch.no_grad()
    def forward(self, points, point_offsets, host_offsets=None):
        feats, coords = self.voxelize_vfe(points, point_offsets, host_offsets)
        canvas = self.sparse_backbone(feats, coords)
        return self.post_process(*self.backbone_head(canvas))
