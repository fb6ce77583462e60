"""Mirror of pcdet/datasets/kitti/kitti_object_eval_python/rotate_iou.py: rotate_iou_gpu_eval with the reference's
signature (numpy in, numpy out), backed by csrc/eval_iou.hip instead of numba.cuda (absent on ROCm)."""
import numpy as np
import torch

from ..... import _lib


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes (N, 5), query_boxes (K, 5): (x, y, w, l, angle).  Returns (N, K) in boxes.dtype (rotate_iou.py:290-330)."""
    box_dtype = boxes.dtype
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float32)
    N, K = boxes.shape[0], query_boxes.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=np.float32).astype(box_dtype)
    dev = torch.device("cuda", device_id)
    with torch.cuda.device(dev):
        b = torch.from_numpy(boxes.reshape(-1, 5)).to(dev)
        q = torch.from_numpy(query_boxes.reshape(-1, 5)).to(dev)
        iou = torch.empty((N, K), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().lidar_rotate_iou_eval(_lib.ptr(b), N, _lib.ptr(q), K, int(criterion), _lib.ptr(iou), _lib.stream()),
                   "lidar_rotate_iou_eval")
        return iou.cpu().numpy().astype(box_dtype)
