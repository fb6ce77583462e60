"""Drop-in replacements for the reference's compiled extension modules.

Each sub-module exposes exactly the functions the reference's pybind11 module of the same name
exports, with the same argument order and the same caller-allocates-outputs
convention, implemented over the C ABI of liblidar_hip.so.  INTEGRATION.md shows the one-line import
change per `*_utils.py` that rebinds the reference onto them.
"""
