"""The helper of pcdet/utils/box_utils.py that the operator layer imports: enlarge_box3d (:136-149)."""
from . import common_utils


def enlarge_box3d(boxes3d, extra_width=(0, 0, 0)):
    """boxes3d (N, 7) [x, y, z, dx, dy, dz, heading]; extra_width: per-axis [w, l, h] or a scalar-like list."""
    boxes3d, is_numpy = common_utils.check_numpy_to_torch(boxes3d)
    large_boxes3d = boxes3d.clone()
    large_boxes3d[:, 3:6] += boxes3d.new_tensor(extra_width)[None, :]
    return large_boxes3d
