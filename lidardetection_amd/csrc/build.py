"""Builds lidardetection_amd/csrc/liblidar_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

-ffp-contract=off: the integer-producing geometry (voxel cells, IoU > thresh, d2 < r2) must evaluate the
reference's fp32 expressions without fused multiply-adds; throughput code uses explicit fmaf().
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liblidar_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]
# per-source extras.  wino43_conv.hip: its 72-slot chunk body must unroll completely (the accumulators are indexed by the slot) and is
# larger than the default budget of `#pragma unroll` (the loop then stays rolled WITHOUT a diagnostic and the accumulators go to
# scratch); the m0 clobber of its LDS-DMA assembly is deliberate
EXTRA_FLAGS = {"wino43_conv.hip": ["-mllvm", "-pragma-unroll-threshold=4000000", "-Wno-inline-asm"]}


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def build(force=False, verbose=False):
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(HERE, "*.h")) + [os.path.abspath(__file__)]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return SO
    objs = []
    procs = []
    for s in srcs:
        o = s[:-4] + ".o"
        if (not force and os.path.exists(o)
                and all(os.path.getmtime(o) >= os.path.getmtime(d) for d in [s] + deps[len(srcs):])):
            objs.append(o)
            continue
        cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(s), []), "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


# A/B variants of the library, loaded by tests through `_lib.load_variant` (never by the product path): name -> (source, extra flags)
VARIANTS = {
    # every bin workgroup of the voxeliser's LDS-binned path behaves as if its shared keys never arrived and takes the
    # deadline exit (s_slow -> vxl_bin_streaming): the branch an ordinary run reaches only beside a stalled neighbour
    "vxl_nowait": ("voxelize.hip", ["-DVXL_WAIT_TICKS=0"]),
}
# timing A/B libraries (tools only, built on request: build_variants(names=[...]))
AB_VARIANTS = {
    "w43_aux0": ("wino43_conv.hip", ["-DW43_STORE_AUX=0"]),
    "vxl_fill_nt": ("voxelize.hip", ["-DVXL_FILL_NT=1"]),          # the voxeliser's zero fill as streaming stores
    "vxl_share12": ("voxelize.hip", ["-DVXL_FILL_SHARE16=12"]), "vxl_share10": ("voxelize.hip", ["-DVXL_FILL_SHARE16=10"]),
    "vxl_share8": ("voxelize.hip", ["-DVXL_FILL_SHARE16=8"]),      # part of the zero fill moved into the emit launch
    "dc_nowait": ("deconv_gemm.hip", ["-DDC_PROBE=1"]),
    "sc_p1": ("sparse_conv.hip", ["-DSC_PROBE=1"]), "sc_p2": ("sparse_conv.hip", ["-DSC_PROBE=2"]), "sc_p4": ("sparse_conv.hip", ["-DSC_PROBE=4"]),
    "sc_p7": ("sparse_conv.hip", ["-DSC_PROBE=7"]),                # WRONG results: ablations of the sparse implicit GEMM            # WRONG results: the deblock GEMM's barrier does not wait for its DMA        # F(4x4) Winograd with default-policy output stores instead of nt
}


def variant_path(name):
    return os.path.join(HERE, f"liblidar_hip_{name}.so")


def build_variants(force=False, verbose=False, names=None):
    build(force=False, verbose=verbose)
    outs = []
    todo = VARIANTS if names is None else {n: {**VARIANTS, **AB_VARIANTS}[n] for n in names}
    for name, (src, extra) in todo.items():
        so = variant_path(name)
        s = os.path.join(HERE, src)
        deps = [s] + glob.glob(os.path.join(HERE, "*.h")) + [os.path.abspath(__file__), SO]
        if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
            outs.append(so)
            continue
        o = os.path.join(HERE, f"{src[:-4]}.{name}.o")
        cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), *extra, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs = [x[:-4] + ".o" for x in sources() if os.path.basename(x) != src] + [o]
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs])
        outs.append(so)
    return outs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_variants(force="--force" in sys.argv, verbose=True))
