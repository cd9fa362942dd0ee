#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
: > $O/ssim_an.txt
for rep in 1 2; do
ABL_TAG=3wg_per_cu timeout -k 10 100 python tools/ssim_timing.py 2>&1 | grep -v amdgpu.ids >> $O/ssim_an.txt
VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_ssim2.so ABL_TAG=2wg_per_cu_forward timeout -k 10 100 python tools/ssim_timing.py 2>&1 | grep -v amdgpu.ids >> $O/ssim_an.txt
done
cat $O/ssim_an.txt
