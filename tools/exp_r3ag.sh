#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_frame.py tests/test_band_loss_gpu.py tests/test_gpu_planned_bins.py -q -m gpu > $O/pytest_ag1.log 2>&1 || { tail -40 $O/pytest_ag1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_ag1.log
: > $O/timing_ag.txt
for rep in 1 2; do
  ABL_TAG=full timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ag.txt 2>&1 || fail full
  ABL_BAND=3/8 ABL_TAG=band3of8 timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ag.txt 2>&1 || fail band
done
ABL_N=5000000 ABL_W=1752 ABL_H=1168 ABL_BAND=3/8 ABL_TAG=scannetpp_5M_band3of8 timeout -k 10 300 python tools/kernel_timing.py >> $O/timing_ag.txt 2>&1 || fail 5m
ABL_N=2000000 ABL_W=640 ABL_H=480 ABL_BAND=1/4 ABL_TAG=scannet_2M_band1of4 timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_ag.txt 2>&1 || fail 2m
grep -v amdgpu.ids $O/timing_ag.txt
