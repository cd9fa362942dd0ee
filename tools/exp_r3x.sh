#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "run_ahead or runs_ahead or growing or graph or overflow" > $O/pytest_x1.log 2>&1 || { tail -30 $O/pytest_x1.log | cut -c1-300; echo FAILED tests; exit 1; }
tail -2 $O/pytest_x1.log
: > $O/host_x.txt
timeout -k 10 200 python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids | head -24 >> $O/host_x.txt
for rep in 1 2; do
ABL_N=10000 ABL_W=320 ABL_H=240 ABL_TAG=cfgA timeout -k 10 100 python tools/kernel_timing.py 2>&1 | grep -v amdgpu.ids >> $O/host_x.txt
ABL_N=400000 ABL_W=640 ABL_H=480 ABL_TAG=tum timeout -k 10 100 python tools/kernel_timing.py 2>&1 | grep -v amdgpu.ids >> $O/host_x.txt
done
cat $O/host_x.txt
