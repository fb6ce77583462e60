"""Grow-only scratch buffers (torch caching allocator, one per (tag, device, STREAM)).

The C ABI never allocates; the host side hands it workspaces.  Buffers are reused across calls so
steady-state launches do no allocation at all (and stay hipGraph-capturable).  A buffer belongs to the stream that was
current when it was requested: two streams running the same operator (e.g. NMS) never share scratch memory."""
import torch

from . import _lib

_BUFS = {}


def get(tag, nbytes, device):
    key = (tag, str(device), int(_lib.stream().value or 0) if torch.device(device).type == "cuda" else 0)
    buf = _BUFS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _BUFS[key] = buf
    return buf


def drop(tag=None):
    for k in [k for k in _BUFS if tag is None or k[0] == tag]:
        del _BUFS[k]
