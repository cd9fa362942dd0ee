#!/bin/bash
# batch AD: how much does the backward composite depend on occupancy?  (3 waves per SIMD shipped; 2 with 32 KB of dummy LDS)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
: > $O/timing_ad.txt
for rep in 1 2; do
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs.so ABL_TAG=3waves timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ad.txt 2>&1 || exit 1
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_bwd2w.so ABL_TAG=2waves timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ad.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $O/timing_ad.txt
