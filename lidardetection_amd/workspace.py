"""Grow-only scratch buffers (torch caching allocator, one per (tag, device)).

The C ABI never allocates; the host side hands it workspaces.  Buffers are reused across calls so
steady-state launches do no allocation at all (and stay hipGraph-capturable)."""
import torch

_BUFS = {}


def get(tag, nbytes, device):
    key = (tag, str(device))
    buf = _BUFS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _BUFS[key] = buf
    return buf


def drop(tag=None):
    for k in [k for k in _BUFS if tag is None or k[0] == tag]:
        del _BUFS[k]
