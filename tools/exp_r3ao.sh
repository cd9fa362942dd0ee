#!/bin/bash
# batch AO: the fused counting sort touches the records of its list entries (L2 warm-up) -- A/B on one box
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "quadrant or sort or bit_identical or variants" > $O/pytest_ao1.log 2>&1 || { tail -30 $O/pytest_ao1.log | cut -c1-300; fail tests; }
tail -2 $O/pytest_ao1.log
: > $O/timing_ao.txt
for rep in 1 2 3; do
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs.so ABL_TAG=touch timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ao.txt 2>&1 || fail touch
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_notouch.so ABL_TAG=notouch timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ao.txt 2>&1 || fail notouch
done
VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs.so ABL_N=500000 ABL_TAG=touch_500k timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ao.txt 2>&1
VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_notouch.so ABL_N=500000 ABL_TAG=notouch_500k timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ao.txt 2>&1
grep -v amdgpu.ids $O/timing_ao.txt
