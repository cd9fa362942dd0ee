#!/bin/bash
# r3g: full GPU suite after the frombuffer fix; graph-replayed iterations; host profile with the C++ node; cfg-A
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
L=$R/vtgaussian-slam_amd/lib
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/pytest_g_full.log 2>&1; tail -6 $O/pytest_g_full.log | cut -c1-600
ABL_N=10000 ABL_W=320 ABL_H=240 ABL_TAG=cfgA-ext python tools/kernel_timing.py > $O/cfga_g.txt 2>&1; tail -2 $O/cfga_g.txt | cut -c1-400
ABL_N=10000 ABL_W=320 ABL_H=240 VTGS_TORCH_EXT=0 ABL_TAG=cfgA-pynode python tools/kernel_timing.py 2>&1 | tail -1 | cut -c1-400 | tee -a $O/cfga_g.txt
ABL_N=10000 ABL_W=320 ABL_H=240 VTGS_TORCH_EXT=0 VTGS_FORWARD_MODE=checked ABL_TAG=cfgA-pynode-checked python tools/kernel_timing.py 2>&1 | tail -1 | cut -c1-400 | tee -a $O/cfga_g.txt
python tools/host_overhead.py > $O/host_g.txt 2>&1; head -32 $O/host_g.txt | cut -c1-200
timeout -k 10 300 python bench_slam.py --frames 3 --get-loss --graph > $O/slam_g1.json 2> $O/slam_g1.err; cut -c1-900 $O/slam_g1.json; tail -3 $O/slam_g1.err
timeout -k 10 300 python bench_slam.py --frames 3 --get-loss > $O/slam_g2.json 2> $O/slam_g2.err; cut -c1-700 $O/slam_g2.json
timeout -k 10 300 python bench_slam.py --frames 2 --get-loss --graph --global-submaps 2 > $O/slam_g3.json 2> $O/slam_g3.err; cut -c1-900 $O/slam_g3.json; tail -2 $O/slam_g3.err
python tools/fused_host_profile.py > $O/fused_host_g.txt 2>&1; tail -12 $O/fused_host_g.txt | cut -c1-200
