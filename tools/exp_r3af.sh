#!/bin/bash
# batch AF: the dense-map BASELINE shapes on the final build (2 M at 640x480 whole + one band of 4; 5 M at 1752x1168 one band of 8)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
: > $O/timing_af.txt
ABL_N=2000000 ABL_W=640 ABL_H=480 ABL_TAG=scannet_2M timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_af.txt 2>&1 || exit 1
ABL_N=2000000 ABL_W=640 ABL_H=480 ABL_BAND=1/4 ABL_TAG=scannet_2M_band1of4 timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_af.txt 2>&1 || exit 1
ABL_N=300000 ABL_W=640 ABL_H=480 ABL_TAG=tum_300k timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_af.txt 2>&1 || exit 1
ABL_N=500000 ABL_TAG=replica_500k timeout -k 10 200 python tools/kernel_timing.py >> $O/timing_af.txt 2>&1 || exit 1
ABL_N=5000000 ABL_W=1752 ABL_H=1168 ABL_BAND=3/8 ABL_TAG=scannetpp_5M_band3of8 timeout -k 10 300 python tools/kernel_timing.py >> $O/timing_af.txt 2>&1 || exit 1
grep -v amdgpu.ids $O/timing_af.txt
