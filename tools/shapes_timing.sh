#!/bin/bash
# The five BASELINE shapes through tools/kernel_timing.py on one box (wall clock per forward+backward step of the drop-in
# operator, HIP-event kernel averages in braces).   bash tools/shapes_timing.sh > gpurun_out/<round>/shapes.log
set -o pipefail
run() { echo "== $1"; shift; env "$@" python tools/kernel_timing.py 2>&1 | tail -1; }
run "cfg-A 10 k @ 320x240"                 ABL_N=10000   ABL_W=320  ABL_H=240  ABL_TAG=cfgA
run "TUM-like 300 k @ 640x480"             ABL_N=300000  ABL_W=640  ABL_H=480  ABL_TAG=tum
run "Replica room0 500 k @ 1200x680"       ABL_N=500000  ABL_W=1200 ABL_H=680  ABL_TAG=replica500k
run "headline 1 M @ 1200x680"              ABL_N=1000000 ABL_W=1200 ABL_H=680  ABL_TAG=headline
run "ScanNet-like 2 M @ 640x480"           ABL_N=2000000 ABL_W=640  ABL_H=480  ABL_TAG=scannet
run "ScanNet-like 2 M @ 640x480, band 1/4" ABL_N=2000000 ABL_W=640  ABL_H=480  ABL_BAND=1/4 ABL_TAG=scannet_band
run "ScanNet++ 5 M @ 1752x1168"            ABL_N=5000000 ABL_W=1752 ABL_H=1168 ABL_TAG=scannetpp
run "ScanNet++ 5 M @ 1752x1168, band 3/8"  ABL_N=5000000 ABL_W=1752 ABL_H=1168 ABL_BAND=3/8 ABL_TAG=scannetpp_band
