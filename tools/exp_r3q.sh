#!/bin/bash
# batch Q: project_and_bin compiled per mode -- whole frame, bands, planned bins
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_planned_bins.py tests/test_gpu_parity.py -q -m gpu -k "planned or band or binning" > $O/pytest_q1.log 2>&1 || { tail -40 $O/pytest_q1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_q1.log
: > $O/timing_q.txt
for rep in 1 2; do
  ABL_TAG=full timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_q.txt 2>&1 || fail full
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_neither.so ABL_TAG=neither timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_q.txt 2>&1 || fail neither
  ABL_BAND=3/8 ABL_TAG=band3of8 timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_q.txt 2>&1 || fail band
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_neither.so ABL_BAND=3/8 ABL_TAG=band3of8_neither timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_q.txt 2>&1 || fail band_neither
  VTGS_BINS=planned ABL_TAG=planned timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_q.txt 2>&1 || fail planned
done
ABL_N=5000000 ABL_W=1752 ABL_H=1168 ABL_BAND=3/8 ABL_TAG=5M_band3of8 timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_q.txt 2>&1 || fail band5m
VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_neither.so ABL_N=5000000 ABL_W=1752 ABL_H=1168 ABL_BAND=3/8 ABL_TAG=5M_band3of8_neither timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_q.txt 2>&1 || fail band5m_neither
grep -v amdgpu.ids $O/timing_q.txt
