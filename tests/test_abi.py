"""The C-ABI library loads and exports every function include/vtgs.h declares; argument validation that
needs no GPU.  (No compute calls here.)"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "vtgs.h")
LIB = os.path.join(ROOT, "vtgaussian-slam_amd", "lib", "libvtgs.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        pytest.fail("libvtgs.so missing: run `python vtgaussian-slam_amd/build.py` (build() does)")
    return ctypes.CDLL(LIB)


def _declared():
    src = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(vtgs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert {"vtgs_forward", "vtgs_backward", "vtgs_forward_shared", "vtgs_mark_visible", "vtgs_workspace_bytes",
            "vtgs_backward_scratch_bytes", "vtgs_strerror", "vtgs_abi_version", "vtgs_last_hip_error",
            "vtgs_debug_layout", "vtgs_profile_enable", "vtgs_profile_collect"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vtgs.h but not exported"


def test_version_strings_and_sizes(lib):
    lib.vtgs_abi_version.restype = ctypes.c_uint32
    assert lib.vtgs_abi_version() == 16
    lib.vtgs_strerror.restype = ctypes.c_char_p
    assert lib.vtgs_strerror(0) == b"ok" and b"instance" in lib.vtgs_strerror(3)
    lib.vtgs_workspace_bytes.restype = ctypes.c_size_t
    lib.vtgs_workspace_bytes.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_uint64, ctypes.c_uint32]
    a = lib.vtgs_workspace_bytes(1000, 640, 480, 8000, 256)
    b = lib.vtgs_workspace_bytes(1000, 640, 480, 8000, 512)
    c = lib.vtgs_workspace_bytes(2000, 640, 480, 8000, 256)
    assert 0 < a < b and a < c and a % 256 == 0
    assert b - a >= 80 * 60 * 256 * 20                                   # 20 bytes per bin slot, 80x60 tiles
    assert lib.vtgs_workspace_bytes(-1, 640, 480, 8, 64) == 0 and lib.vtgs_workspace_bytes(10, 0, 480, 8, 64) == 0
    assert lib.vtgs_workspace_bytes(10, 640, 480, 8, 0) == 0
    # the head of the workspace a forward needs zeroed (VTGS_FORWARD_WORKSPACE_CLEARED): the 256-byte counters block + one
    # list length per 8x8 tile (+ 1), padded to 256 -- and it lies inside every workspace
    lib.vtgs_workspace_clear_bytes.restype = ctypes.c_size_t
    lib.vtgs_workspace_clear_bytes.argtypes = [ctypes.c_int32] * 2
    assert lib.vtgs_workspace_clear_bytes(1200, 680) == 256 + (((150 * 85 + 1) * 4 + 255) // 256) * 256
    assert lib.vtgs_workspace_clear_bytes(640, 480) < lib.vtgs_workspace_bytes(0, 640, 480, 8, 64)
    assert lib.vtgs_workspace_clear_bytes(0, 480) == 0 and lib.vtgs_workspace_clear_bytes(640, -1) == 0
    lib.vtgs_backward_scratch_bytes.restype = ctypes.c_size_t
    lib.vtgs_backward_scratch_bytes.argtypes = [ctypes.c_int32, ctypes.c_uint64]
    assert lib.vtgs_backward_scratch_bytes(10, 100) >= 100 * 40                 # 10-float records


def test_invalid_arguments_are_rejected_before_any_device_work(lib):
    assert lib.vtgs_forward(None, 0, *([None] * 9), ctypes.c_size_t(0), ctypes.c_uint64(1), 64, None, 0, None) == 1
    assert lib.vtgs_mark_visible(None, 0, None, None, None) == 1
    out = (ctypes.c_uint64 * 12)()
    lib.vtgs_debug_layout.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
    assert lib.vtgs_debug_layout(5, 33, 17, 64, 32, out) == 0
    assert out[7] == 5 * 3 and out[3] == 256            # ceil(33/8) x ceil(17/8) tiles; tile_cnt right after counters


def test_python_surface_matches_reference_call_sites():
    import inspect

    import diff_gaussian_rasterization as dgr
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered")                              # utils/recon_helpers.py:14-26
    sig = inspect.signature(dgr.GaussianRasterizer.forward)
    assert list(sig.parameters)[1:] == ["means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                        "rotations", "cov3D_precomp"]
    import torch
    st = dgr.GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4)[None], torch.eye(4)[None],
                                           0, torch.zeros(3), False)
    r = dgr.GaussianRasterizer(raster_settings=st)
    z = torch.zeros(2, 3)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(z, z, z[:, :1], scales=z, rotations=torch.zeros(2, 4))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(z, z, z[:, :1], colors_precomp=z)
    with pytest.raises(RuntimeError, match="no CPU path"):                 # product path never falls back to CPU
        r(z, z, z[:, :1], colors_precomp=z, scales=z, rotations=torch.zeros(2, 4))
    with pytest.raises(ValueError):
        dgr.GaussianRasterizer(raster_settings=st, radius_rule="bogus")


def test_fused_operators_refuse_cpu_tensors():
    """render_frame, the loss node, FusedAdam and the pose reduction are HIP-only: CPU tensors raise, nothing falls back."""
    import torch
    from diff_gaussian_rasterization import losses
    from diff_gaussian_rasterization.optim import FusedAdam
    from diff_gaussian_rasterization.partition import pose7_reduce
    img = torch.rand(3, 8, 8, requires_grad=True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        losses.tracking_loss(img, torch.rand(3, 8, 8), torch.rand(3, 8, 8), torch.rand(1, 8, 8), 0.99)
    with pytest.raises(RuntimeError, match="no CPU path"):
        losses.fused_ssim(img, torch.rand(3, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU path"):
        pose7_reduce(torch.rand(4, 3), torch.rand(4, 3))
    p = torch.nn.Parameter(torch.rand(4, 3))
    p.grad = torch.rand(4, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        FusedAdam([{"params": [p], "lr": 1e-3}]).step()


def test_option_api_without_a_gpu(lib):
    """vtgs_set_option / vtgs_get_option: known names round-trip, value < 0 restores the default, unknown names are
    rejected (no HIP call involved)."""
    lib.vtgs_set_option.restype, lib.vtgs_set_option.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]
    lib.vtgs_get_option.restype, lib.vtgs_get_option.argtypes = ctypes.c_int, [ctypes.c_char_p]
    for name in (b"VTGS_FWD_IMPL", b"VTGS_BWD_IMPL", b"VTGS_BIN_IMPL", b"VTGS_SORT_PACKED", b"VTGS_SORT_FUSED", b"VTGS_COUNT_STEPS"):
        dflt = lib.vtgs_get_option(name)
        assert dflt >= 0
        assert lib.vtgs_set_option(name, 0) == 0 and lib.vtgs_get_option(name) == 0
        assert lib.vtgs_set_option(name, -1) == 0 and lib.vtgs_get_option(name) == dflt
    assert lib.vtgs_set_option(b"VTGS_NO_SUCH_SWITCH", 1) == 1 and lib.vtgs_get_option(b"VTGS_NO_SUCH_SWITCH") == -1


def test_header_is_plain_c99_and_links(tmp_path):
    """include/vtgs.h is the boundary a C / cgo / JNI caller binds: it must compile as C99 (no C++, no torch types) and a C
    program must link against libvtgs.so with nothing but the header."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "vtgaussian-slam_amd", "lib")
    src = tmp_path / "abi.c"
    src.write_text('#include "vtgs.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%u %u %u %u %s\\n", vtgs_abi_version(), (unsigned)sizeof(VtgsForwardInfo), '
                   '(unsigned)sizeof(VtgsCamera), (unsigned)sizeof(VtgsProfileEntry), vtgs_strerror(VTGS_ERR_INSTANCE_OVERFLOW)); '
                   'return vtgs_get_option("VTGS_FWD_IMPL") < 0; }\n')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src),
                    "-L", lib_dir, "-lvtgs", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out[0] == "16"
    # the ctypes mirrors of the package describe the same records as the header
    import diff_gaussian_rasterization as dgr
    assert [int(x) for x in out[1:4]] == [ctypes.sizeof(dgr._VtgsForwardInfo), ctypes.sizeof(dgr._VtgsCamera),
                                          ctypes.sizeof(dgr._VtgsProfileEntry)]


def test_planned_bins_entry_points_check_their_arguments(lib):
    """Planned bins (include/vtgs.h): the plan has one offset per 8x8 tile + 1; the PLANNED bit of tile_capacity does not
    change the workspace size; a planned forward without a plan (or a plan without the bit) is refused before any device work."""
    PLANNED = 0x80000000
    lib.vtgs_bin_plan_entries.restype = ctypes.c_uint32
    assert lib.vtgs_bin_plan_entries(33, 17) == 5 * 3 + 1 and lib.vtgs_bin_plan_entries(0, 17) == 0
    assert lib.vtgs_bin_plan_uniform(33, 17, 64, None, None) == 1
    lib.vtgs_workspace_bytes.restype = ctypes.c_size_t
    lib.vtgs_workspace_bytes.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_uint64, ctypes.c_uint32]
    assert lib.vtgs_workspace_bytes(1000, 640, 480, 8000, 256) == lib.vtgs_workspace_bytes(1000, 640, 480, 8000, PLANNED | 256)
    assert lib.vtgs_workspace_bytes(1000, 640, 480, 8000, PLANNED) == 0
    out = (ctypes.c_uint64 * 12)()
    lib.vtgs_debug_layout.argtypes = [ctypes.c_int32] * 3 + [ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
    assert lib.vtgs_debug_layout(5, 33, 17, 64, PLANNED | 32, out) == 0 and out[11] == 15 * 32 and out[10] > out[8]
    args = [None, 0] + [None] * 9 + [ctypes.c_size_t(0), ctypes.c_uint64(1)]
    assert lib.vtgs_forward_planned(*args, ctypes.c_uint32(PLANNED | 64), None, None, 0, None) == 1


def test_cross_check_kernels_live_in_the_test_only_library():
    """libvtgs.so (the product) carries the default composites only; the scalar / quad-form / lane = pixel forward and the
    quadrant-queue backward -- cross-checks for the GPU tests -- are in libvtgs_xcheck.so, which exports its three entry points
    and is opened by libvtgs.so only when an implementation switch asks (csrc/vtgs_xcheck.hip)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "vtgaussian-slam_amd", "lib")
    x = ctypes.CDLL(os.path.join(lib_dir, "libvtgs_xcheck.so"))
    for name in ("vtgs_xcheck_abi_version", "vtgs_xcheck_forward", "vtgs_xcheck_backward"):
        assert hasattr(x, name), name
    x.vtgs_xcheck_abi_version.restype = ctypes.c_uint32
    main = ctypes.CDLL(os.path.join(lib_dir, "libvtgs.so"))
    main.vtgs_abi_version.restype = ctypes.c_uint32
    assert x.vtgs_xcheck_abi_version() == main.vtgs_abi_version()
    host = subprocess.run(["nm", "-C", os.path.join(lib_dir, "libvtgs.so")], capture_output=True, text=True, check=True).stdout
    for kernel in ("composite_forward_mx", "composite_forward_px", "composite_backward_q", "vtgs::composite_forward(", "vtgs::composite_backward("):
        assert kernel not in host, kernel
    assert "composite_forward_q" in host and "composite_backward_mx<4, false, true, false>" in host and "gather_splat_grads" in host
