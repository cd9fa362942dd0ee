"""The entry scripts the driver runs on the GPU box -- bench.py, bench_slam.py, __graft_entry__.py -- and the measurement tools
at least compile here (a syntax error would otherwise only show on the box), and their argument parsers accept what the driver
and the documentation pass."""
import os
import py_compile
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_entry_scripts_and_tools_compile(tmp_path):
    files = [os.path.join(ROOT, f) for f in ("bench.py", "bench_slam.py", "__graft_entry__.py")]
    files += [os.path.join(ROOT, "tools", f) for f in sorted(os.listdir(os.path.join(ROOT, "tools"))) if f.endswith(".py")]
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


@pytest.mark.parametrize("script,args", [
    ("bench.py", ["--gpus", "1", "--steps", "5", "--warmup", "2", "--mode", "mapping", "--band", "3/8", "--backend", "gloo", "--help"]),
    ("bench_slam.py", ["--frames", "2", "--fused", "--backend", "gloo", "--help"]),
])
def test_argument_parsers_know_the_documented_flags(script, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "usage" in r.stdout.lower(), r.stderr[-400:]


def test_bench_slam_parser_defaults():
    sys.path.insert(0, ROOT)
    try:
        import bench_slam
        a = bench_slam.parse_args(["--frames", "3", "--get-loss"])
        assert a.fused and a.get_loss and a.frames == 3 and a.backend == "nccl" and a.global_submaps == 0
        b = bench_slam.parse_args(["--frames", "1", "--fused", "--backend", "gloo"])
        assert b.fused and not b.get_loss and b.backend == "gloo"
    finally:
        sys.path.remove(ROOT)
