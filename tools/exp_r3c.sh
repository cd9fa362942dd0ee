#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
L=$R/vtgaussian-slam_amd/lib
VTGS_LIBRARY=$L/libvtgs_stamps.so python tools/forward_stamps.py 2>&1 | tail -12 | tee $O/stamps_c.txt
