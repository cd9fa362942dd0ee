#!/bin/bash
# batch P: where did project_and_bin's 14 us come from (same box A/B of experiment builds)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
: > $O/timing_p.txt
for rep in 1 2; do
for lib in libvtgs libvtgs_noycull libvtgs_noplan libvtgs_neither; do
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/$lib.so ABL_TAG=$lib timeout -k 10 120 python tools/kernel_timing.py >> $O/timing_p.txt 2>&1 || { tail -5 $O/timing_p.txt; echo FAILED $lib; exit 1; }
done
done
cat $O/timing_p.txt
