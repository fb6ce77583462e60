"""SECOND-MultiHead NuScenes forward + per-class NMS on one MI355X (BASELINE.json configs[4], one DDP rank's share).

Topology from tools/cfgs/nuscenes_models/cbgs_second_multihead.yaml (10 classes, 6 heads) on the NuScenes voxel grid of
tools/cfgs/dataset_configs/nuscenes_dataset.yaml:20,57-80 (0.1 x 0.1 x 0.2 m, 10 points / voxel, <= 60 000 voxels, 5 point
features): batched HIP voxelise -> MeanVFE -> VoxelResBackBone8x (pcdet/models/backbones_3d/spconv_backbone.py:166-261: rulebooks
+ mask-ordered MFMA implicit GEMM, residual blocks fused) -> HeightCompression -> BaseBEVBackbone [5, 5] (folded, HIP epilogues)
-> AnchorHeadMulti (pcdet/models/dense_heads/anchor_head_multi.py:8-250: shared 3x3 conv, per head a class branch and five
separate regression branches reg:2 height:1 size:3 angle:2 velo:2, SEPARATE_MULTIHEAD, no direction classifier) -> ResidualCoder
with encode_angle_by_sincos (pcdet/utils/box_coder_utils.py:45-77) -> MULTI_CLASSES_NMS per head and class: score >= 0.1, top 1000,
rotated NMS 0.2, first 83 (detector3d_template.py:215-235 -> model_nms_utils.py:28-65).

What differs from the reference is the scheduling of the post-processing, not its arithmetic: instead of B x 10 Python-level
`multi_classes_nms` passes (boolean-mask compaction, `nonzero`, one top-k and one NMS call with its host round trip each), the ten
(head, class) score columns of all frames are ranked by ONE batched top-k, only the 1000 survivors per column are decoded
(decoding is per anchor, so the boxes are the same numbers), and ONE batched device NMS (`lidar_nms_batch_limited`) suppresses all
B x 10 lists at once.  Outputs are padded to 83 per (frame, class) with a count.  Random-init weights; synthetic clouds.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import anchor_post, pillar_ops, synth, wino
from .bev_backbone import _WINO, FoldedBEVBackbone, bias_act_, collect_params, params_key
from .bev_backbone import _fold as fold_bn
from .ext import iou3d_nms_cuda
from .pcdet.models.backbones_3d import spconv_backbone
from .pcdet.utils.cfg import AttrDict
from .pointpillar import make_bev_backbone
from .voxelizer import BatchVoxelizer, grid_size_of

# cbgs_second_multihead.yaml:36-147: (class, anchor size dx dy dz, bottom height); rotations [0, 1.57] for every class
NUS_CLASSES = [
    ("car", [4.63, 1.97, 1.74], -0.95), ("truck", [6.93, 2.51, 2.84], -0.6), ("construction_vehicle", [6.37, 2.85, 3.19], -0.225),
    ("bus", [10.5, 2.94, 3.47], -0.085), ("trailer", [12.29, 2.90, 3.87], 0.115), ("barrier", [0.50, 2.53, 0.98], -1.33),
    ("motorcycle", [2.11, 0.77, 1.47], -1.085), ("bicycle", [1.70, 0.60, 1.28], -1.18), ("pedestrian", [0.73, 0.67, 1.77], -0.935),
    ("traffic_cone", [0.41, 0.41, 1.07], -1.285),
]
NUS_HEADS = [["car"], ["truck", "construction_vehicle"], ["bus", "trailer"], ["barrier"], ["motorcycle", "bicycle"],
             ["pedestrian", "traffic_cone"]]                                   # RPN_HEAD_CFGS :150-169
NUS_REG_LIST = [("reg", 2), ("height", 1), ("size", 3), ("angle", 2), ("velo", 2)]   # SEPARATE_REG_CONFIG :171-174
NUS_ROTATIONS = [0.0, 1.57]


def class_anchors(pc_range, feat_hw, size, rotations, bottom):
    """anchors of one class in the multi-head order [size, rot, z, y, x] flattened -> (len(rot) * H * W, 7)
    (anchor_generator.py:17-61 followed by anchor_head_template.py:241-242's permute(3, 4, 0, 1, 2, 5))."""
    H, W = feat_hw
    xs = torch.arange(pc_range[0], pc_range[3] + 1e-5, step=(pc_range[3] - pc_range[0]) / (W - 1), dtype=torch.float32)
    ys = torch.arange(pc_range[1], pc_range[4] + 1e-5, step=(pc_range[4] - pc_range[1]) / (H - 1), dtype=torch.float32)
    zs = torch.tensor([bottom], dtype=torch.float32)
    X, Y, Z = torch.meshgrid([xs, ys, zs], indexing="ij")
    a = torch.stack((X, Y, Z), dim=-1)[:, :, :, None, :]                                          # [x, y, z, 1, 3]
    a = torch.cat((a, torch.tensor(size, dtype=torch.float32).view(1, 1, 1, 1, 3).expand(*a.shape[:3], 1, 3)), dim=-1)
    a = a[:, :, :, :, None, :].repeat(1, 1, 1, 1, len(rotations), 1)
    r = torch.tensor(rotations, dtype=torch.float32).view(1, 1, 1, 1, -1, 1).expand(*a.shape[:3], 1, len(rotations), 1)
    a = torch.cat((a, r), dim=-1).permute(2, 1, 0, 3, 4, 5).contiguous()                          # [z, y, x, size, rot, 7]
    a[..., 2] += a[..., 5] / 2
    return a.permute(3, 4, 0, 1, 2, 5).contiguous().view(-1, 7)                                   # [size, rot, z, y, x]


def decode_sincos(enc, anchors):
    """ResidualCoder(code_size=9, encode_angle_by_sincos=True).decode_torch (box_coder_utils.py:45-77): enc (..., 10) =
    [xt yt zt dxt dyt dzt cos sin vx vy] against 7-value anchors (their zero padding adds nothing to vx, vy) -> (..., 9)."""
    xa, ya, za, dxa, dya, dza, ra = torch.split(anchors, 1, dim=-1)
    xt, yt, zt, dxt, dyt, dzt, cost, sint, vx, vy = torch.split(enc, 1, dim=-1)
    diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
    rg = torch.atan2(sint + torch.sin(ra), cost + torch.cos(ra))
    return torch.cat([xt * diagonal + xa, yt * diagonal + ya, zt * dza + za, torch.exp(dxt) * dxa, torch.exp(dyt) * dya,
                      torch.exp(dzt) * dza, rg, vx + 0.0, vy + 0.0], dim=-1)


class SingleHead(nn.Module):
    """One RPN head of AnchorHeadMulti (anchor_head_multi.py:8-148) with SEPARATE_REG_CONFIG and no direction classifier: its
    BaseBEVBackbone part is empty for this config (RPN_HEAD_CFGS carries no LAYER_NUMS), so forward() is the branches only."""

    def __init__(self, cin, num_class, anchors_per_loc, mid=64):
        super().__init__()
        self.num_class, self.A = num_class, anchors_per_loc

        def branch(cout):
            return nn.Sequential(nn.Conv2d(cin, mid, 3, padding=1, bias=False), nn.BatchNorm2d(mid), nn.ReLU(),
                                 nn.Conv2d(mid, cout, 3, padding=1, bias=True))
        self.conv_cls = branch(anchors_per_loc * num_class)
        self.conv_box = nn.ModuleDict({f"conv_{name}": branch(anchors_per_loc * ch) for name, ch in NUS_REG_LIST})
        self.code_size = sum(ch for _, ch in NUS_REG_LIST)
        nn.init.constant_(self.conv_cls[-1].bias, -np.log((1 - 0.01) / 0.01))

    def forward(self, x):
        """-> cls (B, A*H*W, num_class), box (B, A*H*W, code) in the reference's anchor-major order (:118-126)."""
        B, _, H, W = x.shape
        cls = self.conv_cls(x)
        box = torch.cat([self.conv_box[f"conv_{name}"](x) for name, _ in NUS_REG_LIST], dim=1)
        box = box.view(B, self.A, self.code_size, H, W).permute(0, 1, 3, 4, 2).reshape(B, -1, self.code_size)
        cls = cls.view(B, self.A, self.num_class, H, W).permute(0, 1, 3, 4, 2).reshape(B, -1, self.num_class)
        return cls, box


class SECONDMultiHeadNuScenes(nn.Module):
    resident_voxels = True      # see PointPillarKITTI.resident_voxels

    def __init__(self, batch_size=4, max_voxels=60000, n_max=30000, device="cuda", score_thresh=0.1, nms_thresh=0.2,
                 nms_pre=1000, nms_post=83):
        super().__init__()
        self.B, self.n_max = batch_size, n_max
        self.pc_range, self.voxel_size = synth.NUS_RANGE, synth.NUS_VOXEL
        self.grid = [int(v) for v in grid_size_of(self.voxel_size, self.pc_range)]          # [1024, 1024, 40]
        self.voxelizer = BatchVoxelizer(self.voxel_size, self.pc_range, 10, max_voxels, 5)
        self.backbone3d = spconv_backbone.VoxelResBackBone8x(AttrDict(), 5, self.grid)
        self.blocks, self.deblocks = make_bev_backbone(cin=256, layer_nums=(5, 5), strides=(1, 2), filters=(128, 256),
                                                       up_strides=(1, 2), up_filters=(256, 256))
        self.shared_conv = nn.Sequential(nn.Conv2d(512, 64, 3, padding=1, bias=False), nn.BatchNorm2d(64, eps=1e-3, momentum=0.01),
                                         nn.ReLU())
        names = [c[0] for c in NUS_CLASSES]
        self.rpn_heads = nn.ModuleList([SingleHead(64, len(h), 2 * len(h)) for h in NUS_HEADS])
        self.head_label_indices = [[names.index(n) + 1 for n in h] for h in NUS_HEADS]       # 1-based labels (:166-168)
        self.score_thresh, self.nms_thresh, self.nms_pre, self.nms_post = score_thresh, nms_thresh, nms_pre, nms_post
        self.to(device).eval()
        for mod in (self.blocks, self.deblocks, self.shared_conv, self.rpn_heads):
            mod.to(memory_format=torch.channels_last)       # 2D part only (the sparse weights are 5-D)
        hw = (self.grid[1] // 8, self.grid[0] // 8)
        per_class = {n: class_anchors(self.pc_range, hw, s, NUS_ROTATIONS, b) for n, s, b in NUS_CLASSES}
        self.head_anchors = [torch.cat([per_class[n] for n in h], 0).to(device) for h in NUS_HEADS]   # (A*H*W, 7) per head
        self._vox_out = self.voxelizer.alloc_outputs(batch_size, device)
        self._bev = None

    def randomize_for_bench(self, seed=0):
        """BN statistics perturbed; class logits spread so that a few thousand anchors per class clear SCORE_THRESH (a trained
        net's typical load on the per-class NMS) — the class-branch bias starts at -4.6 (pi = 0.01), i.e. nothing would pass."""
        g = torch.Generator(device="cpu").manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                    m.running_mean.copy_(torch.empty(m.num_features).uniform_(-0.1, 0.1, generator=g))
                    m.running_var.copy_(torch.empty(m.num_features).uniform_(0.8, 1.2, generator=g))
            for i, h in enumerate(self.rpn_heads):           # later heads: fewer anchors above the threshold (ragged counts)
                h.conv_cls[-1].bias.fill_(-0.5 * i)
                h.conv_cls[-1].weight.mul_(3.0)
        self._bev = None
        return self

    # ---- stages --------------------------------------------------------------------------------
    def voxelize_vfe(self, points, point_offsets):
        vox = self.voxelizer(points, point_offsets, self.n_max, compact=True, out=self._vox_out, resident=self.resident_voxels)
        total = int(vox["voxel_offsets"][self.B])               # the sparse stack needs exact row counts (one read-back)
        feats = pillar_ops.mean_vfe(vox["voxels"][:total], vox["voxel_num_points"][:total])
        return feats, vox["voxel_coords"][:total]

    def sparse_backbone(self, feats, coords):
        bd = self.backbone3d({"voxel_features": feats, "voxel_coords": coords, "batch_size": self.B})
        return bd["encoded_spconv_tensor"].dense_bev()          # (B, 128 * 2, 128, 128), channels-last, one pass

    def bev_features(self, canvas):
        if self._bev is None or self._bev.stale():
            self._bev = FoldedBEVBackbone(self.blocks, self.deblocks, [])
        return self._bev.features(canvas)                       # (B, 512, 128, 128) channels-last

    def heads_reference_layout(self, spatial_2d):
        """the module sequence as the reference runs it: shared conv, then 6 heads x 6 branches of conv/BN/ReLU/conv (144 launches)"""
        x = self.shared_conv(spatial_2d)
        return [h(x) for h in self.rpn_heads]                   # [(cls (B, n_h, c_h), box (B, n_h, 10))] per head

    def _folded_heads(self):
        """every branch's first 3x3 convolution reads the same 64-channel map: their 36 weight sets stacked give ONE convolution
        64 -> 2304 (BatchNorm folded, shift + ReLU in one in-place pass: 1.4 ms instead of 2.1 ms + 72 BatchNorm / ReLU launches).
        The 36 second convolutions (64 -> 2..12 channels) stay separate: as one grouped convolution they measured 2.8 ms
        against 1.5 ms (tools/heads_conv_probe.py).  Cached until a parameter changes."""
        srcs = self.__dict__.get("_heads_src")
        if srcs is None:
            srcs = self.__dict__["_heads_src"] = collect_params(self.shared_conv, self.rpn_heads)
        key = params_key(srcs)
        cache = self.__dict__.get("_heads_folded")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                ws, bs = fold_bn(self.shared_conv[0].weight, self.shared_conv[1], 0, self.shared_conv[0].bias)
                w1, b1, second = [], [], []
                for head in self.rpn_heads:
                    for br in [head.conv_cls] + [head.conv_box[f"conv_{name}"] for name, _ in NUS_REG_LIST]:
                        w, b = fold_bn(br[0].weight, br[1], 0, br[0].bias)
                        w1.append(w)
                        b1.append(b)
                        second.append(br[3])
                cl = lambda t: t.contiguous(memory_format=torch.channels_last)
                w1c = torch.cat(w1, 0)
                # both merged 3x3 layers are stride 1 / padding 1: Winograd F(2x2, 3x3) with shift + ReLU in the kernel (csrc/wino_conv.hip)
                pk = None
                if _WINO[0] and ws.is_cuda and wino.supported(ws.shape[1], ws.shape[0]) and wino.supported(w1c.shape[1], w1c.shape[0]):
                    # the 36 second-layer branch convolutions (64 -> 2..12 channels each) as ONE grouped Winograd launch: group g =
                    # branch g, its filters zero-padded to 32 output channels (36 MIOpen launches + zero-fills + bias adds before)
                    mid0 = w1c.shape[0] // len(second)
                    ok2 = all(tuple(c.kernel_size) == (3, 3) and tuple(c.stride) == (1, 1) and tuple(c.padding) == (1, 1)
                              and c.weight.shape[0] <= 32 and c.weight.shape[1] == mid0 for c in second) and wino.supported(mid0, 32 * len(second))
                    pk2 = None
                    if ok2:
                        w2 = ws.new_zeros((32 * len(second), mid0, 3, 3))
                        b2 = ws.new_zeros((32 * len(second),))
                        for g_, c in enumerate(second):
                            w2[32 * g_:32 * g_ + c.weight.shape[0]] = c.weight.detach()
                            if c.bias is not None:
                                b2[32 * g_:32 * g_ + c.weight.shape[0]] = c.bias.detach()
                        pk2 = [wino.pack_weights(w2), b2, mid0, [c.weight.shape[0] for c in second], None]
                    pk = (wino.pack_auto(ws), wino.pack_auto(w1c), pk2)
                cache = (key, cl(ws), bs.contiguous(), cl(w1c), torch.cat(b1, 0).contiguous(), second, pk)
            self.__dict__["_heads_folded"] = cache
        return cache[1:]

    def heads(self, spatial_2d):
        """same outputs as heads_reference_layout(): shared conv and all first-layer branch convolutions merged and folded"""
        ws, bs, w1, b1, second, pk = self._folded_heads()
        if pk is not None and _WINO[0] and spatial_2d.is_contiguous(memory_format=torch.channels_last):
            x = wino.conv3x3_auto(spatial_2d, pk[0], ws.shape[0], bs, True)
            y = wino.conv3x3_auto(x, pk[1], w1.shape[0], b1, True)
        else:
            x = bias_act_(F.conv2d(spatial_2d, ws, None, padding=1), bs, True)
            y = bias_act_(F.conv2d(x, w1, None, padding=1), b1, True)
        mid = w1.shape[0] // len(second)
        if pk is not None and pk[2] is not None and _WINO[0] and y.is_contiguous(memory_format=torch.channels_last):
            p2, b2, mid0, couts, tables = pk[2]
            z_all, pk[2][4] = wino.conv3x3_grouped_compact(y, p2, mid0, couts, b2, False, tables)     # (B, 236, H, W): real channels only
            offs = np.concatenate([[0], np.cumsum(couts)])
            z = [z_all[:, int(offs[g]):int(offs[g + 1])] for g in range(len(second))]
        else:
            z = [conv(y[:, g * mid:(g + 1) * mid]) for g, conv in enumerate(second)]
        B, _, H, W = y.shape
        out, g = [], 0
        for head in self.rpn_heads:
            cls, box = z[g], torch.cat(z[g + 1:g + 1 + len(NUS_REG_LIST)], dim=1)
            g += 1 + len(NUS_REG_LIST)
            box = box.view(B, head.A, head.code_size, H, W).permute(0, 1, 3, 4, 2).reshape(B, -1, head.code_size)
            cls = cls.view(B, head.A, head.num_class, H, W).permute(0, 1, 3, 4, 2).reshape(B, -1, head.num_class)
            out.append((cls, box))
        return out

    def candidates(self, head_out):
        """-> per (head, class) column: sigmoid scores ranked by one top-k, their anchors decoded.
        scores (B, 10, pre), boxes (B, 10, pre, 9), counts (B, 10) i32 (#scores >= SCORE_THRESH among the top `pre`),
        labels (10,) the 1-based class of each column."""
        # all ten (head, class) columns through ONE top-k, one gather of the encodings, one of the anchors and one decode (per
        # column that is ~10 launches x 10 columns; the columns of one-class heads are padded to the longest with -2: below the
        # -1 of a score under SCORE_THRESH, and k <= every column's true length, so padding is never ranked)
        dev = head_out[0][0].device
        B = head_out[0][0].shape[0]
        n_max = max(cls.shape[1] for cls, _ in head_out)
        n_cols = sum(cls.shape[2] for cls, _ in head_out)
        k = min(self.nms_pre, min(cls.shape[1] for cls, _ in head_out))
        masked = torch.full((B, n_cols, n_max), -2.0, dtype=torch.float32, device=dev)
        col_off, lab, c0, off = [], [], 0, 0
        for (cls, box), labels in zip(head_out, self.head_label_indices):
            prob = torch.sigmoid(cls)                                        # detector3d_template.py:210
            n_h, c_h = prob.shape[1], prob.shape[2]
            masked[:, c0:c0 + c_h, :n_h] = torch.where(prob >= self.score_thresh, prob, prob.new_full((), -1.0)).transpose(1, 2)
            col_off += [off] * c_h
            lab += list(labels)
            c0 += c_h
            off += n_h
        if anchor_post.topk_supported(n_max, k, self.score_thresh):                             # csrc/topk.hip: exact, ties by ascending anchor index; slots
            scores, idx, _ = anchor_post.topk_desc(masked.view(B * n_cols, n_max), k, self.score_thresh)   # past the valid ones hold (-1, 0)
            scores, idx = scores.view(B, n_cols, k), idx.view(B, n_cols, k)
        else:
            scores, idx = torch.topk(masked, k, dim=2)                       # sorted descending == nms_gpu's own sort
        gidx = idx + torch.tensor(col_off, device=dev).view(1, -1, 1)        # row in the heads' concatenated anchor order
        boxes_all = torch.cat([box for _, box in head_out], dim=1)           # (B, sum n_h, 10)
        anchors_all = self._anchors_cat()
        enc = torch.gather(boxes_all, 1, gidx.view(B, -1, 1).expand(-1, -1, boxes_all.shape[2])).view(B, n_cols, k, -1)
        boxes = decode_sincos(enc, anchors_all[gidx])
        counts = (scores >= self.score_thresh).sum(-1).to(torch.int32)
        return scores, boxes, counts, torch.tensor(lab, device=dev)

    def _anchors_cat(self):
        if self.__dict__.get("_anchors_all") is None:
            self.__dict__["_anchors_all"] = torch.cat(self.head_anchors, 0)
        return self.__dict__["_anchors_all"]

    def batched_class_nms(self, scores, boxes, counts):
        """all B x 10 candidate lists through ONE batched device NMS -> keep (B, 10, post) positions into the candidate lists
        (only the first num are valid), num (B, 10)"""
        B, K, P = scores.shape
        b7 = boxes[..., :7].reshape(B * K, P, 7).contiguous()
        keep, num = iou3d_nms_cuda.nms_batch(b7, counts.reshape(-1).contiguous(), self.nms_thresh, max_keep=self.nms_post)
        post = min(self.nms_post, P)
        return keep[:, :post].reshape(B, K, post), torch.clamp(num, max=post).reshape(B, K)

    def post_process(self, head_out):
        """-> boxes (B, 10*post, 9), scores (B, 10*post), labels (B, 10*post), valid mask (B, 10*post): class by class in head
        order, as the reference concatenates them (detector3d_template.py:232-235), padded per class."""
        scores, boxes, counts, labels = self.candidates(head_out)
        keep, num = self.batched_class_nms(scores, boxes, counts)
        B, K, post = keep.shape
        valid = torch.arange(post, device=keep.device).view(1, 1, -1) < num.unsqueeze(-1)
        sel = torch.where(valid, keep, torch.zeros_like(keep)).clamp_(0, scores.shape[2] - 1)
        out_boxes = torch.gather(boxes, 2, sel.unsqueeze(-1).expand(-1, -1, -1, boxes.shape[-1]))
        out_scores = torch.gather(scores, 2, sel)
        out_labels = labels.view(1, K, 1).expand(B, K, post)
        return (out_boxes.reshape(B, K * post, -1), out_scores.reshape(B, -1), out_labels.reshape(B, -1), valid.reshape(B, -1))

    @torch.no_grad()
    def forward(self, points, point_offsets):
        feats, coords = self.voxelize_vfe(points, point_offsets)
        canvas = self.sparse_backbone(feats, coords)
        return self.post_process(self.heads(self.bev_features(canvas)))
