"""Builds libvtgs.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python vtgaussian-slam_amd/build.py [--force]

The library lands in vtgaussian-slam_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("vtgs_api.hip", "vtgs_binning.hip", "vtgs_composite.hip", "vtgs_composite_q.hip", "vtgs_composite_bq.hip", "vtgs_frame.hip", "vtgs_loss.hip", "vtgs_p2p.hip")]
HDR = [os.path.join(HERE, "csrc", f) for f in ("vtgs_internal.h", "vtgs_math.h", "vtgs_composite_common.h", "vtgs_sort_common.h")] + \
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


def build(force: bool = False, verbose: bool = False, out: str = OUT, extra=()) -> str:
    """`out` / `extra` (-D switches): experiment builds next to the shipped library (select one with VTGS_LIBRARY)."""
    if out == OUT and not extra and not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [HIPCC] + [f for f in FLAGS if f] + list(extra) + SRC + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = OUT
    if "--out" in args:
        out = os.path.abspath(args[args.index("--out") + 1])
        del args[args.index("--out"):args.index("--out") + 2]
    print(build(force="--force" in sys.argv, verbose=True, out=out, extra=args))
