#!/bin/bash
# A/B of experiment builds on ONE box: tools/ab_timing.sh <rounds> <lib.so|shipped> ...   (interleaved rounds of tools/kernel_timing.py)
R=$1; shift
for r in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = "shipped" ]; then ABL_TAG=shipped python tools/kernel_timing.py 2>&1 | tail -1
    else VTGS_LIBRARY=$PWD/vtgaussian-slam_amd/lib/$lib ABL_TAG=$lib python tools/kernel_timing.py 2>&1 | tail -1; fi
  done
done
