"""Mirror of pcdet/models/model_utils/model_nms_utils.py: class_agnostic_nms (:6-25), multi_classes_nms (:28-65)."""
import torch

from ...ops.iou3d_nms import iou3d_nms_utils


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    src_box_scores = box_scores
    if score_thresh is not None:
        scores_mask = (box_scores >= score_thresh)
        box_scores, box_preds = box_scores[scores_mask], box_preds[scores_mask]
    selected = []
    if box_scores.shape[0] > 0:
        box_scores_nms, indices = torch.topk(box_scores, k=min(nms_config.NMS_PRE_MAXSIZE, box_scores.shape[0]))
        keep_idx, _ = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)(box_preds[indices][:, 0:7], box_scores_nms,
                                                                    nms_config.NMS_THRESH, **nms_config)
        selected = indices[keep_idx[:nms_config.NMS_POST_MAXSIZE]]
    if score_thresh is not None:
        selected = scores_mask.nonzero().view(-1)[selected]
    return selected, src_box_scores[selected]


def multi_classes_nms(cls_scores, box_preds, nms_config, score_thresh=None):
    """cls_scores (N, num_class), box_preds (N, 7+C) -> per-class NMS, concatenated (scores, labels, boxes)."""
    pred_scores, pred_labels, pred_boxes = [], [], []
    for k in range(cls_scores.shape[1]):
        if score_thresh is not None:
            scores_mask = (cls_scores[:, k] >= score_thresh)
            box_scores, cur_box_preds = cls_scores[scores_mask, k], box_preds[scores_mask]
        else:
            box_scores, cur_box_preds = cls_scores[:, k], box_preds
        selected = []
        if box_scores.shape[0] > 0:
            box_scores_nms, indices = torch.topk(box_scores, k=min(nms_config.NMS_PRE_MAXSIZE, box_scores.shape[0]))
            keep_idx, _ = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)(cur_box_preds[indices][:, 0:7], box_scores_nms,
                                                                        nms_config.NMS_THRESH, **nms_config)
            selected = indices[keep_idx[:nms_config.NMS_POST_MAXSIZE]]
        pred_scores.append(box_scores[selected])
        pred_labels.append(box_scores.new_ones(len(selected)).long() * k)
        pred_boxes.append(cur_box_preds[selected])
    return torch.cat(pred_scores, dim=0), torch.cat(pred_labels, dim=0), torch.cat(pred_boxes, dim=0)
