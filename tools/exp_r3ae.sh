#!/bin/bash
# batch AE: how much does the forward composite depend on occupancy?  (4 waves per SIMD shipped; 3 with 12.5 KB of dummy LDS)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
: > $O/timing_ae.txt
for rep in 1 2; do
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs.so ABL_TAG=4waves timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ae.txt 2>&1 || exit 1
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_fwd3w.so ABL_TAG=3waves timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_ae.txt 2>&1 || exit 1
done
grep -v amdgpu.ids $O/timing_ae.txt
