"""lidardetection_amd — MI355X-native (gfx950) implementation of the LiDAR-detection per-frame hot path.

Layers (bottom-up):
  csrc/*.hip         hand-written HIP kernels + the C ABI declared in include/lidar_hip.h
  _lib.py            ctypes binding of liblidar_hip.so (fails loudly when the library is missing)
  ext/               drop-in replacements of the reference's pybind modules (`iou3d_nms_cuda`, ...):
                     same function names and argument meaning, caller-allocated outputs
  pcdet/ops/...      mirror of the reference's Python operator layer (`pcdet.ops.*`)
  spconv/            mirror of the `spconv` symbols the reference imports
There is no CPU fallback: without a GPU + the built library every op raises.
"""
__version__ = "0.1.0"
