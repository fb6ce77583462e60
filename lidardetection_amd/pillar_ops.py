"""Host side of the PillarVFE / MeanVFE / PointPillarScatter kernels (include/lidar_hip.h)."""
import torch

from . import _lib, workspace


def fold_bn(gamma, beta, mean, var, eps):
    """BatchNorm1d (eval) -> per-channel scale/shift: y = x*scale + shift."""
    scale = gamma / torch.sqrt(var + eps)
    return scale.contiguous(), (beta - mean * scale).contiguous()


def pillar_vfe(voxels, num_points, coords, weight, scale, shift, voxel_size, point_cloud_range,
               with_distance=False, num_voxels_dev=None):
    """Fused PillarVFE (single PFN layer, eval).  voxels (V,P,C) f32; num_points (V) i32|f32;
    coords (V,4) [b,z,y,x] i32|f32; weight (cout, C+6[+1]); -> (V, cout) f32.
    Reference: pcdet/models/backbones_3d/vfe/pillar_vfe.py:94-123 + PFNLayer :29-49."""
    _lib.require_cuda(voxels, num_points, coords, weight, scale, shift)
    V, P, C = voxels.shape
    cout = weight.shape[0]
    if weight.shape[1] != C + 6 + int(bool(with_distance)):
        raise _lib.LidarHipError("weight must be (cout, C + 6 [+1 with_distance]) — use_absolute_xyz layout")
    if coords.dtype not in (torch.int32, torch.float32) or num_points.dtype not in (torch.int32, torch.float32):
        raise _lib.LidarHipError("coords / num_points must be int32 or float32")
    out = torch.empty((V, cout), dtype=torch.float32, device=voxels.device)
    L = _lib.lib()
    _lib.check(L.lidar_pillar_vfe(_lib.ptr(voxels), _lib.ptr(num_points), _lib.ptr(coords), V, _lib.ptr(num_voxels_dev),
                                  P, C, _lib.ptr(weight), _lib.ptr(scale), _lib.ptr(shift), cout,
                                  _lib.host_f32(voxel_size), _lib.host_f32(point_cloud_range), int(bool(with_distance)),
                                  int(coords.dtype == torch.float32), int(num_points.dtype == torch.float32),
                                  _lib.ptr(out), _lib.stream()), "lidar_pillar_vfe")
    return out


def mean_vfe(voxels, num_points):
    """MeanVFE (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-31) -> (V, C)."""
    _lib.require_cuda(voxels, num_points)
    V, P, C = voxels.shape
    out = torch.empty((V, C), dtype=torch.float32, device=voxels.device)
    L = _lib.lib()
    _lib.check(L.lidar_mean_vfe(_lib.ptr(voxels), _lib.ptr(num_points), V, P, C, int(num_points.dtype == torch.float32),
                                _lib.ptr(out), _lib.stream()), "lidar_mean_vfe")
    return out


def pillar_scatter(pillar_features, coords, batch_size, nx, ny, num_voxels_dev=None, out=None, channels_last=False):
    """PointPillarScatter (pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37), nz == 1.
    pillar_features (V, C) f32, coords (V,4) [b,z,y,x] i32|f32 -> (B, C, ny, nx) f32."""
    _lib.require_cuda(pillar_features, coords)
    V, C = pillar_features.shape
    if out is None:
        out = torch.empty((batch_size, C, ny, nx), dtype=torch.float32, device=pillar_features.device,
                          memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    L = _lib.lib()
    wsb = L.lidar_pillar_scatter_workspace_bytes(batch_size, nx, ny)
    ws = workspace.get("scatter", wsb, pillar_features.device)
    _lib.check(L.lidar_pillar_scatter(_lib.ptr(pillar_features), _lib.ptr(coords), int(coords.dtype == torch.float32), V,
                                      _lib.ptr(num_voxels_dev), C, batch_size, nx, ny, int(bool(channels_last)), _lib.ptr(out),
                                      _lib.ptr(ws), wsb,
                                      _lib.stream()), "lidar_pillar_scatter")
    return out


class ResidentCanvas:
    """One persistent channels-last BEV canvas (B, C, ny, nx) kept in HBM across calls: update() clears the cells the previous
    call wrote and writes the new pillars (~2*V*C*4 bytes instead of rewriting the > 90 % zero canvas).  After update() the
    canvas equals pillar_scatter(..., channels_last=True) of the same pillars."""

    def __init__(self, batch_size, channels, ny, nx, max_pillars, device):
        self.B, self.C, self.ny, self.nx, self.cap = batch_size, channels, ny, nx, int(max_pillars)
        self.canvas = torch.zeros((batch_size, channels, ny, nx), dtype=torch.float32, device=device).contiguous(
            memory_format=torch.channels_last)
        self.prev_cells = torch.full((max(self.cap, 1),), -1, dtype=torch.int32, device=device)
        self.prev_count = torch.zeros((1,), dtype=torch.int32, device=device)

    def update(self, pillar_features, coords, num_voxels_dev=None):
        _lib.require_cuda(pillar_features, coords)
        V, C = pillar_features.shape
        if C != self.C or V > self.cap:
            raise _lib.LidarHipError("ResidentCanvas.update: feature width / pillar count exceeds what the canvas was built for")
        _lib.check(_lib.lib().lidar_pillar_scatter_update(_lib.ptr(pillar_features), _lib.ptr(coords), int(coords.dtype == torch.float32), V,
                                                          _lib.ptr(num_voxels_dev), C, self.B, self.nx, self.ny, _lib.ptr(self.canvas),
                                                          _lib.ptr(self.prev_cells), _lib.ptr(self.prev_count), _lib.stream()),
                   "lidar_pillar_scatter_update")
        return self.canvas


def pillar_conv_table(coords, batch_size, nx, ny, k, stride, pad, num_voxels_dev=None):
    """Neighbour table of the k x k / stride / pad convolution that follows PointPillarScatter, over all output pixels in NHWC map
    order: (B * OH * OW, k * k) int32 pillar rows or -1 (csrc/pillar.hip lidar_pillar_conv_table).  -> (nbr, OH, OW)"""
    _lib.require_cuda(coords)
    if coords.dtype not in (torch.int32, torch.float32) or coords.dim() != 2 or coords.shape[1] != 4:
        raise _lib.LidarHipError("pillar_conv_table: coords must be (V, 4) [b, z, y, x] int32 or float32")
    OH, OW = (ny + 2 * pad - k) // stride + 1, (nx + 2 * pad - k) // stride + 1
    L = _lib.lib()
    nbr = torch.empty((batch_size * OH * OW, k * k), dtype=torch.int32, device=coords.device)
    wsb = L.lidar_pillar_conv_table_workspace_bytes(batch_size, nx, ny)
    ws = workspace.get("pillar_conv_table", wsb, coords.device)
    _lib.check(L.lidar_pillar_conv_table(_lib.ptr(coords), int(coords.dtype == torch.float32), coords.shape[0], _lib.ptr(num_voxels_dev),
                                         batch_size, nx, ny, int(k), int(stride), int(pad), _lib.ptr(nbr), _lib.ptr(ws), wsb, _lib.stream()),
               "lidar_pillar_conv_table")
    return nbr, OH, OW


class PillarMap:
    """What PointPillarScatter's output IS before anybody materialises it: the PFN rows, their coordinates and the count (on the
    device).  A consumer that can work from the pillars (FoldedBEVBackbone's sparse first layer) never builds the > 90 %-zero
    canvas; dense() gives the ordinary channels-last canvas (through the owner's ResidentCanvas) to everybody else."""

    def __init__(self, features, coords, num_voxels_dev, batch_size, nx, ny, dense_fn):
        self.features, self.coords, self.num_voxels_dev = features, coords, num_voxels_dev
        self.B, self.nx, self.ny, self._dense_fn = batch_size, nx, ny, dense_fn

    def dense(self):
        return self._dense_fn(self.features, self.coords, self.num_voxels_dev)
