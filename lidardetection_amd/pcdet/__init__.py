"""Mirror of the slice of the reference's `pcdet` package that sits on the hot path.

Only the operator layer (`pcdet.ops.*`) and the pure-tensor modules either side of it (VFE, BEV
scatter, NMS helpers, voxel data-processor) are mirrored — same module paths, names, signatures and
error behaviour as /root/reference/pcdet, implemented on liblidar_hip.so.  Everything else of the
reference (detectors, heads, datasets, CLI) is out of scope and is meant to be used unmodified on
top of these ops.
"""
