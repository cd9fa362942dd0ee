#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_planned_bins.py tests/test_gpu_abi_modes.py tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu > $O/pytest_al1.log 2>&1 || { tail -40 $O/pytest_al1.log | cut -c1-300; echo FAILED tests; exit 1; }
tail -2 $O/pytest_al1.log
VTGS_BINS=planned ABL_N=1000000 ABL_W=640 ABL_H=480 ABL_TAG=1M_640x480_planned timeout -k 10 200 python tools/kernel_timing.py 2>&1 | grep -v amdgpu.ids
VTGS_BINS=planned ABL_TAG=headline_planned timeout -k 10 200 python tools/kernel_timing.py 2>&1 | grep -v amdgpu.ids
