#!/bin/bash
# r3b: A/B on one box: round-2 library vs three-chunk ring vs two-chunk ring (explicit invariant)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
L=$R/vtgaussian-slam_amd/lib
for rep in 1 2; do
VTGS_ABI_ANY=1 VTGS_LIBRARY=$L/libvtgs_r2.so ABL_TAG=r2 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_b.txt
ABL_TAG=ring3 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_b.txt
VTGS_LIBRARY=$L/libvtgs_c2.so ABL_TAG=ring2 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_b.txt
done
VTGS_LIBRARY=$L/libvtgs_c2.so python tools/forward_steps.py 2>&1 | tail -1 | tee $O/steps_b_ring2.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "quadrant" > $O/pytest_b1.log 2>&1; tail -3 $O/pytest_b1.log | cut -c1-400
