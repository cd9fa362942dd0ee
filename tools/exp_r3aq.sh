#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -q -m gpu > $O/pytest_aq1.log 2>&1 || { grep -n "AssertionError\|grad " $O/pytest_aq1.log | head -20 | cut -c1-300; tail -5 $O/pytest_aq1.log; echo FAILED; exit 1; }
tail -2 $O/pytest_aq1.log
