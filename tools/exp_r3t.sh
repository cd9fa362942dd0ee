#!/bin/bash
# batch T: integer wave reductions through DPP + readlane -- correctness (sort / binning tests) and A/B of the kernels
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_planned_bins.py tests/test_gpu_configs.py -q -m gpu -x > $O/pytest_t1.log 2>&1 || { tail -40 $O/pytest_t1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_t1.log
: > $O/timing_t.txt
for rep in 1 2 3; do
  ABL_TAG=dpp timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_t.txt 2>&1 || fail dpp
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_nodpp.so ABL_TAG=shuffle timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_t.txt 2>&1 || fail nodpp
done
ABL_BAND=3/8 ABL_TAG=dpp_band timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_t.txt 2>&1
VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_nodpp.so ABL_BAND=3/8 ABL_TAG=shuffle_band timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_t.txt 2>&1
grep -v amdgpu.ids $O/timing_t.txt
