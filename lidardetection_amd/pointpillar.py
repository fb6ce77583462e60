"""PointPillar-KITTI forward + NMS on one MI355X — the measured end-to-end graph (bench.py).

The reference Python tree does not travel to the GPU box, so this file re-states the PointPillar
topology from tools/cfgs/kitti_models/pointpillar.yaml with
  * this repo's HIP ops for the hot path: batched voxelise -> fused PillarVFE -> BEV scatter -> batched
    rotated NMS (device-resident greedy, no host sync), and
  * stock torch.nn (MIOpen) for the dense 2D backbone + anchor head, as BASELINE.json:configs[1] says.
Shapes / semantics follow: BaseBEVBackbone (pcdet/models/backbones_2d/base_bev_backbone.py:6-112),
AnchorHeadSingle (pcdet/models/dense_heads/anchor_head_single.py:8-60), AnchorGenerator
(pcdet/models/dense_heads/target_assigner/anchor_generator.py:17-61), ResidualCoder.decode_torch
(pcdet/utils/box_coder_utils.py:45-77), direction bins (anchor_head_template.py:253-266),
post_processing + class_agnostic_nms (pcdet/models/detectors/detector3d_template.py:169-275,
pcdet/models/model_utils/model_nms_utils.py:6-25).

Post-processing differs from the reference only in being batched and sync-free: the per-sample Python
loop with boolean-mask indexing is replaced by a masked top-k over all frames at once, boxes are
decoded after the top-k (decode is per-anchor, so the selected boxes are identical), and outputs are
padded to NMS_POST_MAXSIZE with a per-frame count.
"""
import numpy as np
import torch
import torch.nn as nn

from . import anchor_post, pillar_ops, synth
from .bev_backbone import FoldedBEVBackbone, collect_params, params_key
from .ext import iou3d_nms_cuda
from .voxelizer import BatchVoxelizer, grid_size_of

KITTI_ANCHORS = [  # pointpillar.yaml:80-110: (size dx,dy,dz), rotations, bottom height
    ([3.9, 1.6, 1.56], [0, 1.57], -1.78),
    ([0.8, 0.6, 1.73], [0, 1.57], -0.6),
    ([1.76, 0.6, 1.73], [0, 1.57], -0.6),
]


def make_bev_backbone(cin=64, layer_nums=(3, 5, 5), strides=(2, 2, 2), filters=(64, 128, 256),
                      up_strides=(1, 2, 4), up_filters=(128, 128, 128)):
    blocks, deblocks = nn.ModuleList(), nn.ModuleList()
    c_in = [cin, *filters[:-1]]
    for i in range(len(layer_nums)):
        layers = [nn.ZeroPad2d(1), nn.Conv2d(c_in[i], filters[i], 3, stride=strides[i], padding=0, bias=False),
                  nn.BatchNorm2d(filters[i], eps=1e-3, momentum=0.01), nn.ReLU()]
        for _ in range(layer_nums[i]):
            layers += [nn.Conv2d(filters[i], filters[i], 3, padding=1, bias=False),
                       nn.BatchNorm2d(filters[i], eps=1e-3, momentum=0.01), nn.ReLU()]
        blocks.append(nn.Sequential(*layers))
        deblocks.append(nn.Sequential(
            nn.ConvTranspose2d(filters[i], up_filters[i], up_strides[i], stride=up_strides[i], bias=False),
            nn.BatchNorm2d(up_filters[i], eps=1e-3, momentum=0.01), nn.ReLU()))
    return blocks, deblocks


def generate_anchors(pc_range, feat_hw, device):
    """-> (H*W*6, 7) anchors in the reference's order [y, x, class, rot] (anchor_generator.py:17-61)."""
    H, W = feat_hw
    out = []
    for size, rots, bottom in KITTI_ANCHORS:
        xs = torch.arange(pc_range[0], pc_range[3] + 1e-5, step=(pc_range[3] - pc_range[0]) / (W - 1), dtype=torch.float32)
        ys = torch.arange(pc_range[1], pc_range[4] + 1e-5, step=(pc_range[4] - pc_range[1]) / (H - 1), dtype=torch.float32)
        zs = torch.tensor([bottom], dtype=torch.float32)
        X, Y, Z = torch.meshgrid([xs, ys, zs], indexing="ij")
        a = torch.stack((X, Y, Z), dim=-1)[:, :, :, None, :]                      # [x,y,z,1,3]
        a = torch.cat((a, torch.tensor(size).view(1, 1, 1, 1, 3).expand(*a.shape[:3], 1, 3)), dim=-1)
        a = a[:, :, :, :, None, :].repeat(1, 1, 1, 1, len(rots), 1)
        r = torch.tensor(rots, dtype=torch.float32).view(1, 1, 1, 1, -1, 1).expand(*a.shape[:3], 1, len(rots), 1)
        a = torch.cat((a, r), dim=-1).permute(2, 1, 0, 3, 4, 5).contiguous()      # [z,y,x,size,rot,7]
        a[..., 2] += a[..., 5] / 2
        out.append(a)
    return torch.cat(out, dim=-3).view(-1, 7).to(device)


def limit_period(val, offset=0.5, period=np.pi):
    return val - torch.floor(val / period + offset) * period


class PointPillarKITTI(nn.Module):
    # the model owns its voxel output buffers and nothing writes to them between forwards: keep their zero padding resident
    # (BatchVoxelizer.__call__(resident=True)); set to False to rewrite the whole padded buffer every call
    resident_voxels = True

    def __init__(self, batch_size=16, max_voxels=16000, n_max=20000, device="cuda",
                 score_thresh=0.1, nms_thresh=0.01, nms_pre=4096, nms_post=500, channels_last=True, fold_bn=True):
        super().__init__()
        self.B, self.n_max = batch_size, n_max
        self.pc_range, self.voxel_size = synth.PP_RANGE, synth.PP_VOXEL
        self.nx, self.ny, _ = [int(v) for v in grid_size_of(self.voxel_size, self.pc_range)]
        self.voxelizer = BatchVoxelizer(self.voxel_size, self.pc_range, 32, max_voxels, 4)
        self.pfn_linear = nn.Linear(10, 64, bias=False)
        self.pfn_norm = nn.BatchNorm1d(64, eps=1e-3, momentum=0.01)
        self.blocks, self.deblocks = make_bev_backbone()
        self.num_class, self.num_anchor_per_loc, self.num_dir_bins = 3, 6, 2
        self.conv_cls = nn.Conv2d(384, self.num_anchor_per_loc * self.num_class, 1)
        self.conv_box = nn.Conv2d(384, self.num_anchor_per_loc * 7, 1)
        self.conv_dir_cls = nn.Conv2d(384, self.num_anchor_per_loc * self.num_dir_bins, 1)
        self.dir_offset, self.dir_limit_offset = 0.78539, 0.0
        self.score_thresh, self.nms_thresh, self.nms_pre, self.nms_post = score_thresh, nms_thresh, nms_pre, nms_post
        self.channels_last = bool(channels_last) and torch.device(device).type == "cuda"
        self.to(device).eval()
        if self.channels_last:   # NHWC strides: MIOpen's fp32 kernels then run without layout transposes (stock torch)
            self.to(memory_format=torch.channels_last)
        self.anchors = generate_anchors(self.pc_range, (self.ny // 2, self.nx // 2), device)
        self._vox_out = self.voxelizer.alloc_outputs(batch_size, device)
        self._folded = None
        self._canvas = None
        self.fold_bn = bool(fold_bn) and self.channels_last   # folded BN + HIP epilogue (bev_backbone.py)
        self._bev = None

    def randomize_for_bench(self, seed=0):
        """Random-init weights; BN running stats and the class bias are perturbed so that every frame
        produces >= NMS_PRE_MAXSIZE candidates above SCORE_THRESH (a trained net's typical load)."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                    m.running_mean.copy_(torch.empty(m.num_features).uniform_(-0.1, 0.1, generator=g))
                    m.running_var.copy_(torch.empty(m.num_features).uniform_(0.8, 1.2, generator=g))
            self.conv_cls.bias.zero_()
            self.conv_cls.weight.mul_(4.0)
        self._folded = self._bev = None
        return self

    def _pfn_folded(self):
        """folded PFN weights, rebuilt whenever a source parameter / BN statistic changed (load_state_dict, training step)"""
        if getattr(self, "_pfn_srcs", None) is None:
            self._pfn_srcs = collect_params(self.pfn_linear, self.pfn_norm)
        key = params_key(self._pfn_srcs)
        if self._folded is None or self._folded[0] != key:
            n = self.pfn_norm
            s, t = pillar_ops.fold_bn(n.weight.detach(), n.bias.detach(), n.running_mean, n.running_var, n.eps)
            self._folded = (key, (self.pfn_linear.weight.detach().contiguous(), s, t))
        return self._folded[1]

    def _bev_folded(self):
        if self._bev is None or self._bev.stale():
            self._bev = FoldedBEVBackbone(self.blocks, self.deblocks, [self.conv_cls, self.conv_box, self.conv_dir_cls])
        return self._bev

    # ---- stages (kept separate so bench.py can time them) ------------------------------------
    def voxelize(self, points, point_offsets, host_offsets=None, timer=None):
        """host_offsets: the same offsets on the host, when the caller has them (kernel arguments instead of a dependent load)"""
        return self.voxelizer(points, point_offsets, self.n_max, compact=True, out=self._vox_out, resident=self.resident_voxels,
                              host_offsets=host_offsets, timer=timer)

    def vfe_scatter(self, vox):
        w, s, t = self._pfn_folded()
        total = vox["voxel_offsets"][self.B:self.B + 1]
        feat = pillar_ops.pillar_vfe(vox["voxels"], vox["voxel_num_points"], vox["voxel_coords"], w, s, t,
                                     self.voxel_size, self.pc_range, num_voxels_dev=total)
        if self.channels_last:      # resident canvas: clear last step's cells, write this step's (130 MB instead of 877 MB)
            def dense(f, c, n):
                if self._canvas is None:
                    self._canvas = pillar_ops.ResidentCanvas(self.B, f.shape[1], self.ny, self.nx, vox["voxels"].shape[0], f.device)
                return self._canvas.update(f, c, num_voxels_dev=n)
            if self.fold_bn and self._bev_folded().sparse_first_ok():
                # the folded backbone runs its first layer from the pillars themselves: no canvas at all (bev_backbone.py)
                return pillar_ops.PillarMap(feat, vox["voxel_coords"], total, self.B, self.nx, self.ny, dense)
            return dense(feat, vox["voxel_coords"], total)
        return pillar_ops.pillar_scatter(feat, vox["voxel_coords"], self.B, self.nx, self.ny, num_voxels_dev=total)

    def backbone_head(self, canvas):
        if self.fold_bn:
            return (self._bev_folded().merged(canvas),)  # (B, H, W, 18 + 42 + 12): consumed in place by post_process
        return self.backbone_head_stock(canvas if torch.is_tensor(canvas) else canvas.dense())

    def split_heads(self, head):
        """merged head (B, H, W, C) -> cls (B, N, 3), box (B, N, 7), dir (B, N, 2) as the reference's view() calls give"""
        a = self.num_anchor_per_loc
        cls, box, dirs = torch.split(head, [a * self.num_class, a * 7, a * self.num_dir_bins], dim=-1)
        return (cls.reshape(self.B, -1, self.num_class), box.reshape(self.B, -1, 7), dirs.reshape(self.B, -1, self.num_dir_bins))

    def backbone_head_stock(self, canvas):
        ups, x = [], canvas
        for blk, de in zip(self.blocks, self.deblocks):
            x = blk(x)
            ups.append(de(x))
        x = torch.cat(ups, dim=1)
        cls = self.conv_cls(x).permute(0, 2, 3, 1).reshape(self.B, -1, self.num_class)
        box = self.conv_box(x).permute(0, 2, 3, 1).reshape(self.B, -1, 7)
        dirs = self.conv_dir_cls(x).permute(0, 2, 3, 1).reshape(self.B, -1, self.num_dir_bins)
        return cls, box, dirs

    def decode(self, enc, anchors, dir_logits):
        xa, ya, za, dxa, dya, dza, ra = torch.split(anchors, 1, dim=-1)
        xt, yt, zt, dxt, dyt, dzt, rt = torch.split(enc, 1, dim=-1)
        diag = torch.sqrt(dxa ** 2 + dya ** 2)
        boxes = torch.cat([xt * diag + xa, yt * diag + ya, zt * dza + za, torch.exp(dxt) * dxa, torch.exp(dyt) * dya,
                           torch.exp(dzt) * dza, rt + ra], dim=-1)
        period = 2 * np.pi / self.num_dir_bins
        dir_labels = torch.max(dir_logits, dim=-1)[1]
        rot = limit_period(boxes[..., 6] - self.dir_offset, self.dir_limit_offset, period)
        boxes[..., 6] = rot + self.dir_offset + period * dir_labels.to(boxes.dtype)
        return boxes

    def post_process(self, *heads):
        """-> boxes (B, post, 7), scores (B, post), labels (B, post), counts (B); all on the device.
        heads = (merged head,) from the folded backbone (HIP score / decode kernels) or (cls, box, dirs) (torch ops)."""
        if len(heads) == 1:
            return self.post_process_fused(heads[0])
        return self.post_process_torch(*heads)

    def select_topk(self, masked, k):
        """(B, N) masked scores (-1 below SCORE_THRESH) -> the k best per frame, sorted descending: (scores (B, k), anchor indices
        (B, k)).  The HIP selection (csrc/topk.hip) breaks ties by ascending anchor index; torch.topk where it does not apply."""
        if masked.is_cuda and anchor_post.topk_supported(masked.shape[1], k, self.score_thresh):
            return anchor_post.topk_desc(masked, k, self.score_thresh)[:2]
        return torch.topk(masked, k, dim=1)

    def post_process_fused(self, head):
        a = self.num_anchor_per_loc
        n = head.shape[1] * head.shape[2] * a if head.dim() == 4 else head.shape[1] * a
        k = min(self.nms_pre, n)
        if anchor_post.topk_supported(n, k, self.score_thresh, hist=True):   # scores + histogram, collect, finalize: 3 launches, ties by ascending anchor index
            ws = anchor_post.topk_workspace(head.shape[0], n, head.device)
            try:
                masked, labels_all = anchor_post.anchor_scores(head, a, self.num_class, self.score_thresh, cls_off=0, topk_ws=ws)
                top_scores, top_idx, counts = anchor_post.topk_desc(masked, k, self.score_thresh, ws, hist_ready=True)
            except BaseException:
                anchor_post.drop_topk_workspace(ws)       # its histogram may be filled and not consumed
                raise
        else:
            masked, labels_all = anchor_post.anchor_scores(head, a, self.num_class, self.score_thresh, cls_off=0)
            top_scores, top_idx = torch.topk(masked, k, dim=1)        # sorted descending == nms_gpu's sort
            counts = (top_scores >= self.score_thresh).sum(dim=1).to(torch.int32)
        boxes = anchor_post.decode_topk(head, top_idx, self.anchors, a, box_off=a * self.num_class,
                                        dir_off=a * (self.num_class + 7), num_dir_bins=self.num_dir_bins,
                                        dir_offset=self.dir_offset, dir_limit_offset=self.dir_limit_offset)
        post = min(self.nms_post, k)
        keep, num = iou3d_nms_cuda.nms_batch(boxes, counts, self.nms_thresh, max_keep=post)   # survivors past `post` are dropped anyway
        return anchor_post.post_nms_gather(boxes, top_scores, top_idx, labels_all, keep, num, post)   # one launch (was 13 torch kernels)

    def post_process_torch(self, cls, box, dirs):
        scores_all, labels_all = torch.sigmoid(cls).max(dim=-1)
        masked = torch.where(scores_all >= self.score_thresh, scores_all, scores_all.new_full((), -1.0))
        k = min(self.nms_pre, masked.shape[1])
        top_scores, top_idx = torch.topk(masked, k, dim=1)            # sorted descending == nms_gpu's sort
        counts = (top_scores >= self.score_thresh).sum(dim=1).to(torch.int32)
        gi = top_idx.unsqueeze(-1)
        boxes = self.decode(torch.gather(box, 1, gi.expand(-1, -1, 7)), self.anchors[top_idx],
                            torch.gather(dirs, 1, gi.expand(-1, -1, self.num_dir_bins))).contiguous()
        return self._nms_and_gather(boxes, top_scores, top_idx, labels_all, counts, k)

    def _nms_and_gather(self, boxes, top_scores, top_idx, labels_all, counts, k):
        post = min(self.nms_post, k)
        keep, num = iou3d_nms_cuda.nms_batch(boxes, counts, self.nms_thresh, max_keep=post)   # survivors past `post` are dropped anyway
        sel = keep[:, :post].clamp_(0, k - 1)
        num = torch.clamp(num, max=post)
        valid = torch.arange(post, device=sel.device).unsqueeze(0) < num.unsqueeze(1)
        sel = torch.where(valid, sel, torch.zeros_like(sel))
        out_boxes = torch.gather(boxes, 1, sel.unsqueeze(-1).expand(-1, -1, 7))
        out_scores = torch.gather(top_scores, 1, sel)
        out_labels = torch.gather(labels_all, 1, torch.gather(top_idx, 1, sel)).long() + 1
        return out_boxes, out_scores, out_labels, num

    @torch.no_grad()
    def forward(self, points, point_offsets, host_offsets=None):
        vox = self.voxelize(points, point_offsets, host_offsets)
        canvas = self.vfe_scatter(vox)
        return self.post_process(*self.backbone_head(canvas))
