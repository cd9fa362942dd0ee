#!/bin/bash
# HIP API trace + kernel trace of a short SLAM run: which runtime calls does the host make per iteration, and does any of them
# wait for the GPU (round 6: a 40 us bubble in front of loss_backward_kernel in EVERY tracking iteration, although the host's
# iteration is shorter than the GPU's)?
#   (on the GPU box)  bash tools/trace_slam_api.sh <tag> [frames] [extra bench_slam.py arguments]
TAG=${1:-x}; FR=${2:-3}; shift 2
R=$PWD; O=$R/gpurun_out/${VTGS_ROUND:-r6}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/slamapi_$TAG
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/slamapi_$TAG -o run -- python3 $R/bench_slam.py --frames $FR --warmup-frames 1 --get-loss --global-submaps 2 --base-frame-every 60 "$@" > $O/slamapi_$TAG.log 2>&1 || { tail -5 $O/slamapi_$TAG.log; exit 1; }
cd $R
python tools/trace_slam_api.py $O/slamapi_$TAG | tee $O/slamapi_${TAG}.txt
rm -rf $O/slamapi_$TAG
