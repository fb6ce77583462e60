"""The helpers of pcdet/utils/common_utils.py that the operator layer imports
(check_numpy_to_torch :46-49, rotate_points_along_z :66-88, get_voxel_centers :97-113)."""
import numpy as np
import torch


def check_numpy_to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def rotate_points_along_z(points, angle):
    """points (B, N, 3+C), angle (B) -> rotated about +z by `angle`."""
    points, is_numpy = check_numpy_to_torch(points)
    angle, _ = check_numpy_to_torch(angle)
    cosa, sina = torch.cos(angle), torch.sin(angle)
    zeros, ones = angle.new_zeros(points.shape[0]), angle.new_ones(points.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    out = torch.cat((torch.matmul(points[:, :, 0:3], rot), points[:, :, 3:]), dim=-1)
    return out.numpy() if is_numpy else out


def get_voxel_centers(voxel_coords, downsample_times, voxel_size, point_cloud_range):
    """voxel_coords (N, 3) [z, y, x] -> centres (N, 3) [x, y, z]."""
    assert voxel_coords.shape[1] == 3
    centers = voxel_coords[:, [2, 1, 0]].float()
    voxel_size = torch.tensor(voxel_size, device=centers.device).float() * downsample_times
    pc_range = torch.tensor(point_cloud_range[0:3], device=centers.device).float()
    return (centers + 0.5) * voxel_size + pc_range
