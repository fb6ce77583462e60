"""Build recipe for oracle/_ref: the reference's OWN CPU entry points, compiled in place.

TEST INFRASTRUCTURE ONLY.  Nothing under lidardetection_amd/ may import this.

What is built (sources are read where they lie under /root/reference, never copied):

  oracle/_ref/iou3d_nms_cuda.so       <- pcdet/ops/iou3d_nms/src/{iou3d_cpu,iou3d_nms,iou3d_nms_api}.cpp
        gives  boxes_iou_bev_cpu  (iou3d_cpu.cpp:232-252) — the rotated-BEV IoU geometry that is
        textually identical to the CUDA device code (iou3d_nms_kernel.cu:14-234).
  oracle/_ref/roiaware_pool3d_cuda.so <- pcdet/ops/roiaware_pool3d/src/roiaware_pool3d.cpp
        gives  points_in_boxes_cpu (roiaware_pool3d.cpp:143-168).

No stand-in headers or stub sources are written.  The two .cpp files include <cuda.h> /
<cuda_runtime_api.h>; genuine copies of those headers ship in this image inside the triton wheel
(triton/backends/nvidia/include) and are used as they are.  The GPU-side functions in the same
translation units reference CUDA runtime symbols and kernel launchers (defined in .cu files we do
not build); these stay *unresolved* in the shared object.  The loader (oracle/ref_loader.py) imports
the modules with RTLD_LAZY, so the unresolved PLT entries are never bound because only the CPU entry
points are ever called.

The .so files are git-ignored (oracle/_ref/ in .gitignore) but travel to the GPU box with gpurun.
When /root/reference is absent (GPU box) this script does nothing and the prebuilt files are used.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = "/root/reference/pcdet/ops"

TARGETS = {
    "iou3d_nms_cuda": [
        f"{REF}/iou3d_nms/src/iou3d_cpu.cpp",
        f"{REF}/iou3d_nms/src/iou3d_nms.cpp",
        f"{REF}/iou3d_nms/src/iou3d_nms_api.cpp",
    ],
    "roiaware_pool3d_cuda": [
        f"{REF}/roiaware_pool3d/src/roiaware_pool3d.cpp",
    ],
}


def _flags():
    import torch
    from torch.utils import cpp_extension as ce
    import triton  # only to locate the genuine CUDA headers that ship in the wheel

    cuda_inc = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
    if not os.path.exists(os.path.join(cuda_inc, "cuda.h")):
        raise RuntimeError("no genuine cuda.h in this image; oracle/_ref is unbuildable")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], cuda_inc]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cflags = ["-O2", "-fPIC", "-std=c++17", "-w", "-ffp-contract=off",
              f"-D_GLIBCXX_USE_CXX11_ABI={abi}"]
    cflags += [f"-I{p}" for p in inc]
    libdir = ce.library_paths()[0]
    ldflags = [f"-L{libdir}", f"-Wl,-rpath,{libdir}", "-lc10", "-ltorch_cpu", "-ltorch",
               "-ltorch_python"]
    return cflags, ldflags


def build(force=False, verbose=False):
    """Returns the list of built/prebuilt .so paths (empty if nothing is available)."""
    os.makedirs(OUT, exist_ok=True)
    have_ref = os.path.isdir(REF)
    built = []
    for name, srcs in TARGETS.items():
        so = os.path.join(OUT, name + ".so")
        if not have_ref:
            if os.path.exists(so):
                built.append(so)
            continue
        if (not force and os.path.exists(so)
                and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs)):
            built.append(so)
            continue
        cflags, ldflags = _flags()
        cmd = ["g++", "-shared", *cflags, f"-DTORCH_EXTENSION_NAME={name}", *srcs, "-o", so, *ldflags]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        built.append(so)
    return built


if __name__ == "__main__":
    print("\n".join(build(force="--force" in sys.argv, verbose=True)))
