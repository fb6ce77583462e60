"""NMS front-ends with the reference's names and call contracts (pcdet/models/model_utils/model_nms_utils.py:6-65), written
around this repo's device-resident NMS: scores below the threshold are masked instead of removed (no boolean-mask compaction of
the box tensor), one top-k orders the survivors, and the rotated NMS named by `nms_config.NMS_TYPE` runs on that ordering.

Return values are the reference's: indices refer to the ORIGINAL (unfiltered) inputs.
"""
import torch

from ...ops.iou3d_nms import iou3d_nms_utils

_MASKED = -1.0e30      # below any real score: masked entries sort last and are cut off by their count


def _ordered_candidates(scores, score_thresh, pre_maxsize):
    """-> (candidate indices into `scores`, best first; their scores).  Only entries >= score_thresh (if given) qualify."""
    if scores.numel() == 0:
        return scores.new_zeros(0, dtype=torch.long), scores
    ranked = scores if score_thresh is None else torch.where(scores >= score_thresh, scores, scores.new_full((), _MASKED))
    count = scores.numel() if score_thresh is None else int((scores >= score_thresh).sum())
    take = min(int(pre_maxsize), count)
    if take == 0:
        return scores.new_zeros(0, dtype=torch.long), scores[:0]
    top, order = torch.topk(ranked, k=take)
    return order, top


def _suppress(order, top_scores, boxes, nms_config):
    """runs NMS_TYPE on boxes[order] (already sorted by score) and maps the survivors back through `order`"""
    if order.numel() == 0:
        return order
    nms_fn = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)
    survivors, _ = nms_fn(boxes[order][:, :7], top_scores, nms_config.NMS_THRESH, **nms_config)
    return order[survivors[:nms_config.NMS_POST_MAXSIZE]]


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    """box_scores (N,), box_preds (N, 7+C) -> (selected indices into the inputs, their scores)."""
    order, top = _ordered_candidates(box_scores, score_thresh, nms_config.NMS_PRE_MAXSIZE)
    selected = _suppress(order, top, box_preds, nms_config)
    return selected, box_scores[selected]


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """cls_scores (N, num_class), box_preds (N, 7+C) -> (scores, labels, boxes) of the per-class survivors, class by class."""
    kept_scores, kept_labels, kept_boxes = [], [], []
    for label in range(cls_scores.shape[1]):
        column = cls_scores[:, label]
        order, top = _ordered_candidates(column, score_thresh, nms_config.NMS_PRE_MAXSIZE)
        chosen = _suppress(order, top, box_preds, nms_config)
        kept_scores.append(column[chosen])
        kept_labels.append(torch.full((chosen.numel(),), label, dtype=torch.long, device=cls_scores.device))
        kept_boxes.append(box_preds[chosen])
    return torch.cat(kept_scores), torch.cat(kept_labels), torch.cat(kept_boxes)
