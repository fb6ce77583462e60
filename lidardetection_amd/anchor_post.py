"""Host side of csrc/anchor_post.hip: class scores + threshold mask, and box decode of the top-k survivors, read straight
from the merged head output (reference: detector3d_template.py:205-230, model_nms_utils.py:6-10,
anchor_head_template.py:226-273, box_coder_utils.py:45-77)."""
import numpy as np
import torch

from . import _lib


_TOPK_WS = {}


def topk_supported(n, k, valid_min=None, hist=False):
    """lidar_topk_desc takes N % 4 == 0 and k <= 4096, scores and threshold of any sign (r04: order-preserving key).  hist=True: the
    fused score + histogram kernel (lidar_anchor_scores_hist) additionally needs a POSITIVE threshold — a config with SCORE_THRESH 0
    goes to the plain score kernel + topk_desc (or torch.topk) in the callers."""
    return n % 4 == 0 and 0 < k <= 4096 and n < 2 ** 31 and (not hist or valid_min is None or float(valid_min) > 0.0)


def drop_topk_workspace(ws):
    """forget a cached workspace whose histogram may be half-consumed (an exception between anchor_scores(topk_ws=ws) and
    topk_desc(hist_ready=True)): the next call gets a freshly initialised one"""
    for key in [k for k, v in _TOPK_WS.items() if v is ws]:
        del _TOPK_WS[key]


def topk_workspace(batch, n, device):
    """workspace of lidar_topk_desc for (batch, n) scores on the current stream (initialised once per buffer)"""
    key = (str(device), int(batch), int(n), int(_lib.stream().value or 0))
    ws = _TOPK_WS.get(key)
    if ws is None:
        L = _lib.lib()
        nbytes = L.lidar_topk_workspace_bytes(batch, n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _lib.check(L.lidar_topk_workspace_init(_lib.ptr(ws), nbytes, batch, n, _lib.stream()), "lidar_topk_workspace_init")
        _TOPK_WS[key] = ws
    return ws


def topk_desc(scores, k, valid_min, ws=None, hist_ready=False, score_max=1.0):
    """scores (B, N) f32 -> (top_scores (B, k) descending, top_idx (B, k) int64, counts (B,) int32): the k best scores >= valid_min
    of every frame, ties by ascending index; slots past counts[b] hold (-1, 0) for valid_min > 0 and (-inf, 0) otherwise.  valid_min may
    be any float, -inf = no threshold (raw logits: roi_head_template.py:45-99); score_max: an upper bound (shapes the histogram only).
    hist_ready: `ws` was handed to anchor_scores (topk_ws=ws) for these very scores."""
    _lib.require_cuda(scores)
    if scores.dtype != torch.float32 or scores.dim() != 2 or not topk_supported(scores.shape[1], k, valid_min, hist=hist_ready):
        raise _lib.LidarHipError("topk_desc: scores must be float32 (B, N) with N % 4 == 0 and k <= 4096 (valid_min > 0 with hist_ready)")
    B, n = scores.shape
    if ws is None:
        ws, hist_ready = topk_workspace(B, n, scores.device), False
    top_scores = torch.empty((B, k), dtype=torch.float32, device=scores.device)
    top_idx = torch.empty((B, k), dtype=torch.int64, device=scores.device)
    counts = torch.empty((B,), dtype=torch.int32, device=scores.device)
    if not float(score_max) >= float(valid_min):
        raise _lib.LidarHipError("topk_desc: score_max must be >= valid_min")
    _lib.check(_lib.lib().lidar_topk_desc(_lib.ptr(scores), B, n, int(k), float(np.float32(valid_min)), float(np.float32(score_max)),
                                          int(bool(hist_ready)), _lib.ptr(top_scores), _lib.ptr(top_idx), _lib.ptr(counts), _lib.ptr(ws),
                                          ws.numel(), _lib.stream()), "lidar_topk_desc")
    return top_scores, top_idx, counts


def anchor_scores(head, anchors_per_loc, num_class, score_thresh, cls_off=0, topk_ws=None):
    """head (B, H, W, C) or (B, P, C) contiguous fp32 -> scores (B, P*A) f32 (-1 below the threshold), labels (B, P*A) u8.
    topk_ws (topk_workspace(B, P*A, device)): also fills its per-frame score histogram — launch 1 of topk_desc(hist_ready=True)."""
    _lib.require_cuda(head)
    B, C = head.shape[0], head.shape[-1]
    n_loc = head.numel() // C
    scores = torch.empty((B, n_loc // B * anchors_per_loc), dtype=torch.float32, device=head.device)
    labels = torch.empty(scores.shape, dtype=torch.uint8, device=head.device)
    if topk_ws is not None:
        _lib.check(_lib.lib().lidar_anchor_scores_hist(_lib.ptr(head), B, n_loc // B, C, int(cls_off), int(anchors_per_loc), int(num_class),
                                                       float(np.float32(score_thresh)), _lib.ptr(scores), _lib.ptr(labels), _lib.ptr(topk_ws),
                                                       topk_ws.numel(), _lib.stream()), "lidar_anchor_scores_hist")
        return scores, labels
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


def post_nms_gather(boxes, top_scores, top_idx, labels_all, keep, num_keep, post):
    """-> out_boxes (B, post, 7), out_scores (B, post), out_labels (B, post) int64 (class + 1), out_num (B) int32: the first `post`
    NMS survivors of every frame, in one launch (model_nms_utils.py:19-25 + detector3d_template.py:236-262, batched)."""
    _lib.require_cuda(boxes, top_scores, top_idx, labels_all, keep, num_keep, allow=(torch.int64, torch.uint8))
    if (boxes.dtype != torch.float32 or top_scores.dtype != torch.float32 or top_idx.dtype != torch.int64 or labels_all.dtype != torch.uint8
            or keep.dtype != torch.int64 or num_keep.dtype != torch.int32):
        raise _lib.LidarHipError("post_nms_gather: boxes / scores f32, top_idx / keep int64, labels uint8, num_keep int32")
    _lib.require_last(boxes, 7, "boxes")
    B, k = top_scores.shape
    n = labels_all.shape[1]
    post = int(post)
    dev = boxes.device
    out_boxes = torch.empty((B, post, 7), dtype=torch.float32, device=dev)
    out_scores = torch.empty((B, post), dtype=torch.float32, device=dev)
    out_labels = torch.empty((B, post), dtype=torch.int64, device=dev)
    out_num = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().lidar_post_nms_gather(_lib.ptr(boxes), _lib.ptr(top_scores), _lib.ptr(top_idx), _lib.ptr(labels_all), _lib.ptr(keep),
                                                _lib.ptr(num_keep), B, k, n, keep.shape[1], post, _lib.ptr(out_boxes), _lib.ptr(out_scores),
                                                _lib.ptr(out_labels), _lib.ptr(out_num), _lib.stream()), "lidar_post_nms_gather")
    return out_boxes, out_scores, out_labels, out_num
