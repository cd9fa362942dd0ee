#!/bin/bash
# rocprofv3 kernel trace of the SLAM loop run up to a LATE frame of a submap cycle (long tile lists, a densified map), then the
# iterations of the LAST ordinary frame taken apart: tracking and mapping iterations separately -- period, GPU-busy time,
# the largest bubbles and the kernel table (round 6: the slam block's tracking time grows from 0.53 to 1.05 ms per iteration
# over a cycle with densification, 0.75 without; is the late iteration bound by the GPU or by the host?)
#   (on the GPU box)  bash tools/trace_slam_late.sh <tag> [frames] [extra bench_slam.py arguments]
TAG=${1:-x}; FR=${2:-22}; shift 2
R=$PWD; O=$R/gpurun_out/${VTGS_ROUND:-r6}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/slamlate_$TAG
rocprofv3 --kernel-trace --output-format csv -d $O/slamlate_$TAG -o run -- python3 $R/bench_slam.py --frames $FR --warmup-frames 1 --get-loss --global-submaps 2 --base-frame-every 60 "$@" > $O/slamlate_$TAG.log 2>&1 || { tail -5 $O/slamlate_$TAG.log; exit 1; }
cd $R; f=$(find $O/slamlate_$TAG -name "*kernel_trace.csv" | head -1)
python tools/trace_slam_late.py $f | tee $O/slamlate_${TAG}.txt
rm -rf $O/slamlate_$TAG          # (the raw trace is tens of MB; the summary is what travels back)
