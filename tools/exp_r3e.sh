#!/bin/bash
# r3e: quadrant-queue backward, second version (16-wide sweeps, fixes); DPP sort exchanges A/B; get_loss branches
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
L=$R/vtgaussian-slam_amd/lib
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "quadrant_queue_backward or sort or depth_sorted" > $O/pytest_e1.log 2>&1; tail -5 $O/pytest_e1.log | cut -c1-600
for rep in 1 2; do
ABL_TAG=dpp-bwd2 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_e.txt
VTGS_LIBRARY=$L/libvtgs_nodpp.so ABL_TAG=nodpp-bwd2 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_e.txt
VTGS_BWD_IMPL=3 ABL_TAG=dpp-bwd3 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_e.txt
done
VTGS_BWD_IMPL=3 VTGS_LIBRARY=$L/libvtgs_stamps.so python tools/forward_stamps.py 2>&1 | tail -24 | tee $O/stamps_e.txt
timeout -k 10 300 python -m pytest tests/test_get_loss_mirror.py tests/test_get_loss_fixtures.py tests/test_submaps_golden.py -q -m gpu -x > $O/pytest_e2.log 2>&1; tail -5 $O/pytest_e2.log | cut -c1-600
