"""`iou3d_nms_cuda` — same entry points as pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:10-17."""
import torch

from .. import _lib, workspace


def _pairwise(boxes_a, boxes_b, out, mode):
    _lib.require_cuda(boxes_a, boxes_b, out)
    _lib.require_last(boxes_a, 7, "boxes_a")
    _lib.require_last(boxes_b, 7, "boxes_b")
    if out.dtype != torch.float32 or tuple(out.shape) != (boxes_a.shape[0], boxes_b.shape[0]):
        raise _lib.LidarHipError("output must be float32 (N, M)")
    na, nb = boxes_a.shape[0], boxes_b.shape[0]
    L = _lib.lib()
    wsb = L.lidar_iou_workspace_bytes(na, nb)
    ws = workspace.get("iou", wsb, boxes_a.device)
    _lib.check(L.lidar_boxes_pairwise_bev(_lib.ptr(boxes_a), na, _lib.ptr(boxes_b), nb, mode, _lib.ptr(out),
                                          _lib.ptr(ws), wsb, _lib.stream()), "lidar_boxes_pairwise_bev")
    return 1


def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
    """iou3d_nms.cpp:49-68: fills ans_overlap (N, M) with rotated BEV intersection areas."""
    return _pairwise(boxes_a, boxes_b, ans_overlap, 0)


def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
    """iou3d_nms.cpp:70-88: fills ans_iou (N, M) with rotated BEV IoU."""
    return _pairwise(boxes_a, boxes_b, ans_iou, 1)


def nms_batch(boxes, counts, thresh, normal=False, max_keep=None):
    """Batched device-resident NMS (no host sync).  boxes (B, N, 7) sorted by score desc.
    -> keep (B, N) int64 positions, num_keep (B,) int32, both on the device.  max_keep: only the first max_keep survivors
    of a frame are needed (NMS_POST_MAXSIZE): the greedy pass stops there, num_keep is clamped."""
    _lib.require_cuda(boxes, counts)
    _lib.require_last(boxes, 7, "boxes")
    if boxes.dim() != 3 or boxes.dtype != torch.float32 or (counts is not None and counts.dtype != torch.int32):
        raise _lib.LidarHipError("nms_batch: boxes must be float32 (B, N, 7), counts int32 (B,)")
    B, N = boxes.shape[0], boxes.shape[1]
    L = _lib.lib()
    keep = torch.empty((B, max(N, 1)), dtype=torch.int64, device=boxes.device)
    num = torch.empty((B,), dtype=torch.int32, device=boxes.device)
    wsb = L.lidar_nms_workspace_bytes(B, N)
    ws = workspace.get("nms", wsb, boxes.device)
    mk = max(N, 1) if max_keep is None else max(int(max_keep), 1)
    _lib.check(L.lidar_nms_batch_limited(_lib.ptr(boxes), _lib.ptr(counts), B, N, float(thresh), int(normal), mk, _lib.ptr(keep),
                                         _lib.ptr(num), _lib.ptr(ws), wsb, _lib.stream()), "lidar_nms_batch_limited")
    return keep, num


def _nms(boxes, keep, thresh, normal):
    _lib.require_cuda(boxes)
    _lib.require_last(boxes, 7, "boxes")
    if keep.is_cuda or keep.dtype != torch.int64:
        raise _lib.LidarHipError("keep must be a CPU int64 tensor (reference contract, iou3d_nms_utils.py:97)")
    n = boxes.shape[0]
    if n == 0:
        return 0
    k, num = nms_batch(boxes.view(1, n, 7), None, thresh, normal)
    num_out = int(num.item())
    keep[:num_out] = k[0, :num_out].cpu()
    return num_out


def nms_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:90-136: boxes (N,7) cuda sorted by score; keep CPU int64 (N); returns #kept."""
    return _nms(boxes, keep, nms_overlap_thresh, False)


def nms_normal_gpu(boxes, keep, nms_overlap_thresh):
    """iou3d_nms.cpp:139-186 (axis-aligned BEV IoU)."""
    return _nms(boxes, keep, nms_overlap_thresh, True)


def nms_mask_debug(boxes, thresh, normal=False):
    """Test hook: the (N, ceil(N/64)) suppression mask as int64 words (upper-triangular tiles only)."""
    n = boxes.shape[0]
    nms_batch(boxes.view(1, n, 7), None, thresh, normal)
    L = _lib.lib()
    ws = workspace.get("nms", 0, boxes.device)
    off = L.lidar_nms_mask_ptr(_lib.ptr(ws), 1, n) - ws.data_ptr()
    cb = (n + 63) // 64
    return ws[off:off + n * cb * 8].view(torch.int64).view(n, cb).clone()


def boxes_iou_bev_cpu(boxes_a, boxes_b, ans_iou):
    """iou3d_cpu.cpp:232-252 — CPU tensors; fills ans_iou (N, M).  Runs on the host, touches no GPU state."""
    if boxes_a.is_cuda or boxes_b.is_cuda or ans_iou.is_cuda:
        raise _lib.LidarHipError("boxes_iou_bev_cpu takes CPU tensors")
    for t in (boxes_a, boxes_b, ans_iou):
        if not t.is_contiguous():
            raise _lib.LidarHipError("expected a contiguous tensor")   # reference: CHECK_CONTIGUOUS + exit(-1)
        if t.dtype != torch.float32:
            raise _lib.LidarHipError(f"expected float32, got {t.dtype}")
    _lib.require_last(boxes_a, 7, "boxes_a")
    _lib.require_last(boxes_b, 7, "boxes_b")
    _lib.check(_lib.lib().lidar_boxes_iou_bev_cpu(_lib.ptr(boxes_a), boxes_a.shape[0], _lib.ptr(boxes_b), boxes_b.shape[0],
                                                  _lib.ptr(ans_iou)), "lidar_boxes_iou_bev_cpu")
    return 1
