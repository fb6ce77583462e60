"""Mirror of pcdet/ops/iou3d_nms/iou3d_nms_utils.py (same symbols, signatures and return types).

boxes are (N, 7) [x, y, z, dx, dy, dz, heading].  The native module is lidardetection_amd.ext.iou3d_nms_cuda.
"""
import torch

from ...utils import common_utils
from ....ext import iou3d_nms_cuda


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:12-28 — CPU tensors / numpy in, (N, M) rotated BEV IoU out."""
    boxes_a, is_numpy = common_utils.check_numpy_to_torch(boxes_a)
    boxes_b, is_numpy = common_utils.check_numpy_to_torch(boxes_b)
    assert not (boxes_a.is_cuda or boxes_b.is_cuda), 'Only support CPU tensors'
    assert boxes_a.shape[1] == 7 and boxes_b.shape[1] == 7
    ans_iou = boxes_a.new_zeros(torch.Size((boxes_a.shape[0], boxes_b.shape[0])))
    iou3d_nms_cuda.boxes_iou_bev_cpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou.numpy() if is_numpy else ans_iou


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_nms_utils.py:31-45 -> (N, M) cuda float."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    ans_iou = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_iou_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), ans_iou)
    return ans_iou


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:48-81 -> (N, M) 3D IoU = BEV overlap x height overlap / union volume."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    boxes_a_height_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    boxes_a_height_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    boxes_b_height_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    boxes_b_height_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)

    overlaps_bev = torch.zeros((boxes_a.shape[0], boxes_b.shape[0]), dtype=torch.float32, device=boxes_a.device)
    iou3d_nms_cuda.boxes_overlap_bev_gpu(boxes_a.contiguous(), boxes_b.contiguous(), overlaps_bev)

    max_of_min = torch.max(boxes_a_height_min, boxes_b_height_min)
    min_of_max = torch.min(boxes_a_height_max, boxes_b_height_max)
    overlaps_h = torch.clamp(min_of_max - max_of_min, min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.py:84-99 -> (kept indices into `boxes` (cuda int64), None)."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    boxes = boxes[order].contiguous()
    keep = torch.LongTensor(boxes.size(0))
    num_out = iou3d_nms_cuda.nms_gpu(boxes, keep, thresh)
    return order[keep[:num_out].to(boxes.device)].contiguous(), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:102-116 (axis-aligned BEV IoU)."""
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    boxes = boxes[order].contiguous()
    keep = torch.LongTensor(boxes.size(0))
    num_out = iou3d_nms_cuda.nms_normal_gpu(boxes, keep, thresh)
    return order[keep[:num_out].to(boxes.device)].contiguous(), None
