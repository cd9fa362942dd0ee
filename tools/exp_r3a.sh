#!/bin/bash
# r3a: forward ring of three chunks (explicit drain invariant): parity, steps per tile, kernel times
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "quadrant or variants_agree" > $O/pytest_a1.log 2>&1; tail -4 $O/pytest_a1.log | cut -c1-400
python tools/forward_steps.py 2>&1 | tail -1 | tee $O/steps_a.txt
ABL_TAG=r3a python tools/kernel_timing.py 2>&1 | grep step | tee $O/timing_a.txt
ABL_TAG=r3a-again python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_a.txt
python -m pytest tests -q -m gpu -x > $O/pytest_a_full.log 2>&1; tail -4 $O/pytest_a_full.log | cut -c1-400
