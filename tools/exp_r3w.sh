#!/bin/bash
# batch W: stability -- memory soak of the operator, a 12-frame SLAM run, the global-set variant
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
sed -i "s#ROOT='/root/repo'#ROOT='$R'#" tools/soak.py
timeout -k 10 300 python tools/soak.py > $O/soak_w.txt 2>&1 || { tail -5 $O/soak_w.txt; echo FAILED soak; exit 1; }
tail -4 $O/soak_w.txt
timeout -k 10 400 python bench_slam.py --frames 12 --get-loss > $O/slam_w12.json 2> $O/slam_w12.err || { tail -5 $O/slam_w12.err; echo FAILED slam12; exit 1; }
cut -c1-700 $O/slam_w12.json
timeout -k 10 400 python bench_slam.py --frames 3 --get-loss --global-submaps 2 > $O/slam_w_g2.json 2> $O/slam_w_g2.err || { tail -5 $O/slam_w_g2.err; echo FAILED slam g2; exit 1; }
cut -c1-500 $O/slam_w_g2.json
