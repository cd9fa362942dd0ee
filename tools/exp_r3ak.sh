#!/bin/bash
# batch AK: lists of 513..1024 entries pre-sorted ahead of the fusing forward
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_planned_bins.py -q -m gpu > $O/pytest_ak1.log 2>&1 || { tail -40 $O/pytest_ak1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_ak1.log
: > $O/timing_ak.txt
for rep in 1 2; do
ABL_N=1000000 ABL_W=640 ABL_H=480 ABL_TAG=1M_640x480 timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_ak.txt 2>&1 || fail a
VTGS_SORT_LONG_COUNTING=0 ABL_N=1000000 ABL_W=640 ABL_H=480 ABL_TAG=1M_640x480_network timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_ak.txt 2>&1 || fail b
done
ABL_N=5000000 ABL_W=1752 ABL_H=1168 ABL_BAND=3/8 ABL_TAG=5M_band3of8 timeout -k 10 300 python tools/kernel_timing.py >> $O/timing_ak.txt 2>&1 || fail c
VTGS_SORT_LONG_COUNTING=0 ABL_N=5000000 ABL_W=1752 ABL_H=1168 ABL_BAND=3/8 ABL_TAG=5M_band3of8_network timeout -k 10 300 python tools/kernel_timing.py >> $O/timing_ak.txt 2>&1 || fail d
ABL_TAG=headline timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_ak.txt 2>&1 || fail e
grep -v amdgpu.ids $O/timing_ak.txt
