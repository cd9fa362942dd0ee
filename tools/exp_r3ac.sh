#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_planned_bins.py tests/test_gpu_parity.py tests/test_gpu_fused_frame.py -q -m gpu -k "planned or band or binning" > $O/pytest_ac1.log 2>&1 || { tail -40 $O/pytest_ac1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_ac1.log
: > $O/timing_ac.txt
for rep in 1 2; do
  ABL_TAG=uniform timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ac.txt 2>&1 || fail uniform
  VTGS_BINS=planned ABL_TAG=planned timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ac.txt 2>&1 || fail planned
  ABL_BAND=3/8 ABL_TAG=band3of8 timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ac.txt 2>&1 || fail band
done
grep -v amdgpu.ids $O/timing_ac.txt
