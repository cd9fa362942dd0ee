#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m pytest tests/test_band_loss_gpu.py -q -m gpu -k "rccl or two_ranks" > $O/pytest_z1.log 2>&1 || { tail -50 $O/pytest_z1.log | cut -c1-300; echo FAILED tests; exit 1; }
tail -2 $O/pytest_z1.log
