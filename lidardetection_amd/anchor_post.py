"""Host side of csrc/anchor_post.hip: class scores + threshold mask, and box decode of the top-k survivors, read straight
from the merged head output (reference: detector3d_template.py:205-230, model_nms_utils.py:6-10,
anchor_head_template.py:226-273, box_coder_utils.py:45-77)."""
import numpy as np
import torch

from . import _lib


def anchor_scores(head, anchors_per_loc, num_class, score_thresh, cls_off=0):
    """head (B, H, W, C) or (B, P, C) contiguous fp32 -> scores (B, P*A) f32 (-1 below the threshold), labels (B, P*A) u8."""
    _lib.require_cuda(head)
    B, C = head.shape[0], head.shape[-1]
    n_loc = head.numel() // C
    scores = torch.empty((B, n_loc // B * anchors_per_loc), dtype=torch.float32, device=head.device)
    labels = torch.empty(scores.shape, dtype=torch.uint8, device=head.device)
    _lib.check(_lib.lib().lidar_anchor_scores(_lib.ptr(head), n_loc, C, int(cls_off), int(anchors_per_loc), int(num_class),
                                              float(score_thresh), _lib.ptr(scores), _lib.ptr(labels), _lib.stream()),
               "lidar_anchor_scores")
    return scores, labels


def decode_topk(head, top_idx, anchors, anchors_per_loc, box_off, dir_off, num_dir_bins, dir_offset, dir_limit_offset):
    """-> boxes (B, k, 7) of the anchors top_idx (B, k) int64 selects, decoded as generate_predicted_boxes does."""
    _lib.require_cuda(head, top_idx, anchors, allow=(torch.int64,))
    _lib.require_last(anchors, 7, "anchors")
    if top_idx.dtype != torch.int64:
        raise _lib.LidarHipError("decode_topk: top_idx must be int64 (torch.topk indices)")
    B, C = head.shape[0], head.shape[-1]
    locs = head.numel() // C // B
    k = top_idx.shape[1]
    boxes = torch.empty((B, k, 7), dtype=torch.float32, device=head.device)
    period = float(np.float32(2 * np.pi / num_dir_bins)) if num_dir_bins else 1.0
    _lib.check(_lib.lib().lidar_decode_topk(_lib.ptr(head), B, locs, C, int(box_off), int(dir_off), int(anchors_per_loc),
                                            int(num_dir_bins), _lib.ptr(top_idx), k, _lib.ptr(anchors), float(np.float32(dir_offset)),
                                            float(np.float32(dir_limit_offset)), period, _lib.ptr(boxes), _lib.stream()),
               "lidar_decode_topk")
    return boxes
