#!/bin/bash
# r3f: backward with the next chunk's records in flight + up-front prologue; C++ autograd node; host-bound shapes; SLAM loop
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
L=$R/vtgaussian-slam_amd/lib
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/pytest_f_full.log 2>&1; tail -6 $O/pytest_f_full.log | cut -c1-600
for rep in 1 2; do
ABL_TAG=ext python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_f.txt
VTGS_TORCH_EXT=0 ABL_TAG=pynode python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_f.txt
VTGS_ABI_ANY=1 VTGS_LIBRARY=$L/libvtgs_r2.so VTGS_FORWARD_MODE=checked ABL_TAG=r2lib python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_f.txt
done
for shape in "10000 320 240 cfgA" "300000 640 480 tum" "500000 1200 680 replica500k"; do
set -- $shape
ABL_N=$1 ABL_W=$2 ABL_H=$3 ABL_TAG=$4-ext python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_f.txt
ABL_N=$1 ABL_W=$2 ABL_H=$3 VTGS_TORCH_EXT=0 ABL_TAG=$4-pynode python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_f.txt
ABL_N=$1 ABL_W=$2 ABL_H=$3 VTGS_TORCH_EXT=0 VTGS_FORWARD_MODE=checked ABL_TAG=$4-pynode-checked python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_f.txt
done
python tools/host_overhead.py 2>&1 | grep "host floor" | tee $O/host_f.txt
VTGS_TORCH_EXT=0 python tools/host_overhead.py 2>&1 | grep "host floor" | tee -a $O/host_f.txt
VTGS_LIBRARY=$L/libvtgs_stamps.so python tools/forward_stamps.py 2>&1 | tail -8 | tee $O/stamps_f.txt
timeout -k 10 300 python bench_slam.py --frames 3 --get-loss > $O/slam_f1.json 2> $O/slam_f1.err; cut -c1-700 $O/slam_f1.json
VTGS_FORWARD_MODE=checked timeout -k 10 300 python bench_slam.py --frames 3 --get-loss > $O/slam_f2.json 2> $O/slam_f2.err; cut -c1-700 $O/slam_f2.json
timeout -k 10 300 python bench_slam.py --frames 2 --get-loss --global-submaps 2 > $O/slam_f3.json 2> $O/slam_f3.err; cut -c1-900 $O/slam_f3.json; tail -2 $O/slam_f3.err
