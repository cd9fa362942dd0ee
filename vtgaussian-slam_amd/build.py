"""Builds libvtgs.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python vtgaussian-slam_amd/build.py [--force]

The library lands in vtgaussian-slam_amd/lib/ (git-ignored, shipped to the GPU box by gpurun), with the test-only
libvtgs_xcheck.so (cross-check implementations of the composites) next to it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("vtgs_api.hip", "vtgs_binning.hip", "vtgs_composite.hip", "vtgs_composite_q.hip", "vtgs_frame.hip", "vtgs_loss.hip", "vtgs_p2p.hip", "vtgs_sh.hip")]
# the cross-check composites (scalar, quad form, lane = pixel forward, quadrant-queue backward): a TEST-ONLY library next to the
# product, built from the same sources with -DVTGS_XCHECK_BUILD=1; libvtgs.so opens it when an implementation switch asks
XCHECK_SRC = [os.path.join(HERE, "csrc", f) for f in ("vtgs_xcheck.hip", "vtgs_composite.hip", "vtgs_composite_bq.hip")]
HDR = [os.path.join(HERE, "csrc", f) for f in ("vtgs_internal.h", "vtgs_math.h", "vtgs_composite_common.h", "vtgs_sort_common.h")] + \
      [os.path.join(HERE, "..", "include", "vtgs.h")]
OUT = os.path.join(HERE, "lib", "libvtgs.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", 
         "-Wall", "-Wno-unused-function"]


def xcheck_path(out: str) -> str:
    return out[:-3] + "_xcheck.so"


def up_to_date() -> bool:
    if not os.path.exists(OUT) or not os.path.exists(xcheck_path(OUT)):
        return False
    t = min(os.path.getmtime(OUT), os.path.getmtime(xcheck_path(OUT)))
    return all(os.path.getmtime(f) <= t for f in SRC + XCHECK_SRC + HDR + [os.path.abspath(__file__)])


def build(force: bool = False, verbose: bool = False, out: str = OUT, extra=()) -> str:
    """`out` / `extra` (-D switches): experiment builds next to the shipped library (select one with VTGS_LIBRARY)."""
    if out == OUT and not extra and not force and up_to_date():
        return OUT
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [HIPCC] + [f for f in FLAGS if f] + list(extra) + SRC + ["-ldl", "-o", out]
    xcmd = [HIPCC] + [f for f in FLAGS if f] + list(extra) + ["-DVTGS_XCHECK_BUILD=1"] + XCHECK_SRC + ["-o", xcheck_path(out)]
    if verbose:
        print(" ".join(cmd))
        print(" ".join(xcmd))
    procs = [subprocess.Popen(c) for c in (cmd, xcmd)]               # the two libraries share no object: build them side by side
    codes = [p.wait() for p in procs]
    if any(codes):
        raise subprocess.CalledProcessError(max(codes), cmd if codes[0] else xcmd)
    return out


TORCH_EXT_SRC = os.path.join(HERE, "csrc", "vtgs_torch.cpp")
TORCH_EXT_OUT = os.path.join(HERE, "lib", "vtgs_torch.so")


def build_torch_ext(force: bool = False, verbose: bool = False) -> str:
    """The C++ autograd node of the operator (csrc/vtgs_torch.cpp: host plumbing, no device code) as an in-tree Python
    extension module, linked against libvtgs.so next to it.  g++ through torch.utils.cpp_extension (ninja)."""
    hdr = os.path.join(HERE, "..", "include", "vtgs.h")
    if (not force and os.path.exists(TORCH_EXT_OUT)
            and os.path.getmtime(TORCH_EXT_OUT) >= max(os.path.getmtime(TORCH_EXT_SRC), os.path.getmtime(hdr))):   # (libvtgs.so is
        # bound at load time through the C ABI: a kernel rebuild does not stale the node)
        return TORCH_EXT_OUT
    from torch.utils import cpp_extension
    lib_dir = os.path.dirname(OUT)
    build_dir = os.path.join(lib_dir, "build_vtgs_torch")
    os.makedirs(build_dir, exist_ok=True)
    cpp_extension.load(name="vtgs_torch", sources=[TORCH_EXT_SRC], extra_cflags=["-O2", "-std=c++17"],
                       extra_ldflags=["-L" + lib_dir, "-lvtgs", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + lib_dir],
                       build_directory=build_dir, verbose=verbose, is_python_module=False)
    import shutil
    shutil.copy2(os.path.join(build_dir, "vtgs_torch.so"), TORCH_EXT_OUT)
    return TORCH_EXT_OUT


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = OUT
    if "--out" in args:
        out = os.path.abspath(args[args.index("--out") + 1])
        del args[args.index("--out"):args.index("--out") + 2]
    print(build(force="--force" in sys.argv, verbose=True, out=out, extra=args))
    if out == OUT:
        print(build_torch_ext(force="--force" in sys.argv, verbose=True))
