#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_planned_bins.py tests/test_gpu_abi_modes.py -q -m gpu > $O/pytest_y1.log 2>&1 || { tail -50 $O/pytest_y1.log | cut -c1-300; echo FAILED tests; exit 1; }
tail -2 $O/pytest_y1.log
