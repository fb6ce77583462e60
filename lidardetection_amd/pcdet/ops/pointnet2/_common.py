"""Pieces shared by the stacked-batch and dense-batch PointNet++ front-ends: the shared-MLP builder, the pooling over a
ball's samples, inverse-distance weights, and small allocation helpers."""
import torch
import torch.nn as nn


def zeros_i32(shape, device):
    return torch.zeros(shape, dtype=torch.int32, device=device)


def empty_i32(shape, device):
    return torch.empty(shape, dtype=torch.int32, device=device)


def empty_f32(shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


def zeros_f32(shape, device):
    return torch.zeros(shape, dtype=torch.float32, device=device)


def require_contiguous(*tensors):
    for t in tensors:
        if not t.is_contiguous():
            raise AssertionError('expected contiguous tensors')


def shared_mlp(widths):
    """1x1 Conv2d + BatchNorm2d + ReLU per consecutive width pair (the reference's layer naming: Sequential indices 0,1,2,...)"""
    stack = []
    for w_in, w_out in zip(widths[:-1], widths[1:]):
        stack.extend((nn.Conv2d(w_in, w_out, kernel_size=1, bias=False), nn.BatchNorm2d(w_out), nn.ReLU()))
    return nn.Sequential(*stack)


def pool_over_samples(x, method):
    """x (..., nsample) -> (...): reduce the last axis"""
    if method == 'max_pool':
        return x.amax(dim=-1)
    if method == 'avg_pool':
        return x.mean(dim=-1)
    raise NotImplementedError(method)


def inverse_distance_weights(dist, eps=1e-8):
    inv = 1.0 / (dist + eps)
    return inv / inv.sum(dim=-1, keepdim=True)


def addmm_act(shift, x, w, relu=True):
    """x @ w + shift (, ReLU): one library GEMM, with the bias + ReLU epilogue fused when torch exposes it"""
    if relu:
        fused = getattr(torch, "_addmm_activation", None)
        if fused is not None:
            return fused(shift, x, w)
        return torch.addmm(shift, x, w).relu_()
    return torch.addmm(shift, x, w)
