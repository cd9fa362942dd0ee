#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "long_list or sort" > $O/pytest_aj1.log 2>&1 || { tail -40 $O/pytest_aj1.log | cut -c1-300; echo FAILED tests; exit 1; }
tail -2 $O/pytest_aj1.log
