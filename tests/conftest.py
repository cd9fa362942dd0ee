import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vtgaussian-slam_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hostsim():
    """ctypes handle on the g++ build of csrc/vtgs_math.h (test infrastructure)."""
    import ctypes
    d = os.path.join(ROOT, "tests", "hostsim")
    so, src = os.path.join(d, "libhostsim.so"), os.path.join(d, "hostsim.cpp")
    hdr = os.path.join(PKG, "csrc", "vtgs_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.hostsim_min_quadratic.restype = ctypes.c_float
    lib.hostsim_min_quadratic.argtypes = [ctypes.c_float] * 9
    return lib


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _restore_library_options():
    """Implementation switches set through vtgs_set_option live in the library, not in the environment: put the
    defaults back after every test (only when the package is already loaded -- CPU-only tests never load it here)."""
    yield
    mod = sys.modules.get("diff_gaussian_rasterization")
    if mod is not None and hasattr(mod, "reset_options"):
        mod.reset_options()
