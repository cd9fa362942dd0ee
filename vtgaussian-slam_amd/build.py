"""Builds libvtgs.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python vtgaussian-slam_amd/build.py [--force]

The library lands in vtgaussian-slam_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("vtgs_api.hip", "vtgs_binning.hip", "vtgs_composite.hip", "vtgs_frame.hip", "vtgs_loss.hip")]
HDR = [os.path.join(HERE, "csrc", f) for f in ("vtgs_internal.h", "vtgs_math.h")] + \
      [os.path.join(HERE, "..", "include", "vtgs.h")]
OUT = os.path.join(HERE, "lib", "libvtgs.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", 
         "-Wall", "-Wno-unused-function"]


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(f) <= t for f in SRC + HDR + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [HIPCC] + [f for f in FLAGS if f] + SRC + ["-o", OUT]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
