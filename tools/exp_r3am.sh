#!/bin/bash
# batch AM: the forward keeps its fused sort + finalize for every bin size; sort_long_lists / sort_tiles only pre-sort
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_planned_bins.py tests/test_gpu_abi_modes.py tests/test_gpu_fused_frame.py -q -m gpu > $O/pytest_am1.log 2>&1 || { tail -40 $O/pytest_am1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_am1.log
: > $O/timing_am.txt
for rep in 1 2; do
ABL_N=2000000 ABL_W=640 ABL_H=480 ABL_TAG=scannet_2M timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_am.txt 2>&1 || fail a
ABL_N=2000000 ABL_W=640 ABL_H=480 ABL_BAND=1/4 ABL_TAG=scannet_2M_band1of4 timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_am.txt 2>&1 || fail b
done
ABL_N=1000000 ABL_W=640 ABL_H=480 ABL_TAG=1M_640x480 timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_am.txt 2>&1 || fail c
ABL_TAG=headline timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_am.txt 2>&1 || fail d
VTGS_BINS=planned ABL_TAG=headline_planned timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_am.txt 2>&1 || fail e
grep -v amdgpu.ids $O/timing_am.txt
