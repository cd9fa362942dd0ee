#!/bin/bash
# batch V: planned bins with the sort fused (long-list pass ahead of the composite)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_planned_bins.py tests/test_gpu_parity.py tests/test_gpu_abi_modes.py -q -m gpu > $O/pytest_v1.log 2>&1 || { tail -40 $O/pytest_v1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_v1.log
: > $O/timing_v.txt
for rep in 1 2; do
  ABL_TAG=uniform timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_v.txt 2>&1 || fail uniform
  VTGS_BINS=planned ABL_TAG=planned timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_v.txt 2>&1 || fail planned
done
grep -v amdgpu.ids $O/timing_v.txt
