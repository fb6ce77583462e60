"""Loads oracle/_ref/*.so (the reference's own CPU code, see build_ref.py).  TEST INFRASTRUCTURE ONLY.

The modules carry unresolved CUDA symbols (they were built from the reference's unmodified
translation units without the .cu files), therefore they are dlopen'ed with RTLD_LAZY: the PLT
entries of the GPU entry points are never bound because they are never called.
"""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE = {}


def load(name):
    """name in {'iou3d_nms_cuda', 'roiaware_pool3d_cuda'} -> module or None if not built."""
    if name in _CACHE:
        return _CACHE[name]
    so = os.path.join(_HERE, "_ref", name + ".so")
    if not os.path.exists(so):
        _CACHE[name] = None
        return None
    import torch  # noqa: F401  (libtorch must be resident before the extension is opened)
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY | os.RTLD_LOCAL)
    try:
        spec = importlib.util.spec_from_file_location(name, so)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.setdlopenflags(old)
    _CACHE[name] = mod
    return mod
