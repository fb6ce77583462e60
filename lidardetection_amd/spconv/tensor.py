"""SparseConvTensor (spconv/__init__.py upstream; ctor call site pcdet/models/backbones_3d/spconv_backbone.py:132-137)."""
import torch

from .. import _lib, workspace


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        """features (N, C) f32; indices (N, 4) int32 [b, z, y, x]; spatial_shape [D, H, W]."""
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid

    @property
    def spatial_size(self):
        n = 1
        for s in self.spatial_shape:
            n *= s
        return n

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def dense(self, channels_first=True):
        """(N, C) rows -> dense (B, C, D, H, W) [channels_first] or (B, D, H, W, C); zeros elsewhere."""
        feats = self.features.contiguous()
        N, C = feats.shape
        B, (D, H, W) = self.batch_size, self.spatial_shape
        if C in (32, 64, 128) and feats.dtype == torch.float32 and feats.is_cuda and not feats.requires_grad:
            L = _lib.lib()
            out = torch.empty((B, C, D, H, W), dtype=torch.float32, device=feats.device)
            wsb = L.lidar_sparse_to_dense_workspace_bytes(B, D, H, W)
            ws = workspace.get("dense", wsb, feats.device)
            idx = self.indices.int().contiguous()
            _lib.check(L.lidar_sparse_to_dense(_lib.ptr(feats), _lib.ptr(idx), N, C, B, D, H, W, _lib.ptr(out), _lib.ptr(ws), wsb,
                                               _lib.stream()), "lidar_sparse_to_dense")
            return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()
        # differentiable / odd-width path: index_put on a zero volume (stock torch)
        out = feats.new_zeros((B, D, H, W, C))
        idx = self.indices.long()
        out[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] = feats
        return out.permute(0, 4, 1, 2, 3).contiguous() if channels_first else out

    def dense_bev(self):
        """HeightCompression in one pass: (B, C*D, H, W) with channels-last strides, equal to
        dense().view(B, C*D, H, W).contiguous(memory_format=torch.channels_last) (height_compression.py:21-24)."""
        feats = self.features.contiguous()
        N, C = feats.shape
        B, (D, H, W) = self.batch_size, self.spatial_shape
        if not (feats.is_cuda and feats.dtype == torch.float32 and C % 4 == 0 and D <= 4 and not feats.requires_grad):
            d = self.dense()
            return d.view(B, C * D, H, W).contiguous(memory_format=torch.channels_last)
        L = _lib.lib()
        out = torch.empty((B, C * D, H, W), dtype=torch.float32, device=feats.device, memory_format=torch.channels_last)
        wsb = L.lidar_sparse_to_dense_workspace_bytes(B, D, H, W)
        ws = workspace.get("dense", wsb, feats.device)
        idx = self.indices.int().contiguous()
        _lib.check(L.lidar_sparse_to_bev_nhwc(_lib.ptr(feats), _lib.ptr(idx), N, C, B, D, H, W, _lib.ptr(out), _lib.ptr(ws), wsb,
                                              _lib.stream()), "lidar_sparse_to_bev_nhwc")
        return out

    @property
    def sparity(self):
        return self.indices.shape[0] / self.spatial_size / self.batch_size
