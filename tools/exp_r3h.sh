#!/bin/bash
# r3h: after the short-list fix: small scenes first, then the full suite, cfg-A, counting sort timing, graphs last.  Stops at the first failure.
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
L=$R/vtgaussian-slam_amd/lib
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "parity or cfg_a or odd_shapes or sort or quadrant" > $O/pytest_h1.log 2>&1 || { tail -15 $O/pytest_h1.log | cut -c1-300; fail "parity subset"; }
tail -2 $O/pytest_h1.log
timeout -k 10 900 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_parity.py::test_forward_backward_replayed_from_a_graph > $O/pytest_h_full.log 2>&1 || { tail -15 $O/pytest_h_full.log | cut -c1-300; fail "full suite"; }
tail -2 $O/pytest_h_full.log
for v in "" "VTGS_TORCH_EXT=0" "VTGS_TORCH_EXT=0 VTGS_FORWARD_MODE=checked"; do
  env $v ABL_N=10000 ABL_W=320 ABL_H=240 ABL_TAG="cfgA $v" python tools/kernel_timing.py > $O/cfga_h.tmp 2>&1 || { tail -5 $O/cfga_h.tmp; fail "cfgA $v"; }
  grep step $O/cfga_h.tmp | tee -a $O/cfga_h.txt
done
for rep in 1 2; do
ABL_TAG=countsort python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_h.txt
VTGS_ABI_ANY=1 VTGS_LIBRARY=$L/libvtgs_r2.so VTGS_FORWARD_MODE=checked ABL_TAG=r2lib python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_h.txt
done
VTGS_LIBRARY=$L/libvtgs_stamps.so python tools/forward_stamps.py 2>&1 | grep -v "^backward" | head -8 | tee $O/stamps_h.txt
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "replayed_from_a_graph" > $O/pytest_h_graph.log 2>&1 || { tail -25 $O/pytest_h_graph.log | cut -c1-300; fail "graph test"; }
tail -2 $O/pytest_h_graph.log
timeout -k 10 300 python bench_slam.py --frames 3 --get-loss --graph > $O/slam_h1.json 2> $O/slam_h1.err || { tail -5 $O/slam_h1.err | cut -c1-300; fail "slam graph"; }
cut -c1-900 $O/slam_h1.json
