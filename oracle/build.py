"""Builds oracle/_build/liblidar_oracle.so from oracle/src/*.c (gcc, -ffp-contract=off).

TEST INFRASTRUCTURE ONLY: the oracle is the checker, never the thing shipped or measured.
"""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "liblidar_oracle.so")


def build(force=False):
    srcs = sorted(glob.glob(os.path.join(HERE, "src", "*.c")))
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    if (not force and os.path.exists(SO)
            and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in srcs)):
        return SO
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
           "-Wall", "-o", SO, *srcs, "-lm"]
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True))
