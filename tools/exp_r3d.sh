#!/bin/bash
# r3d: quadrant-queue backward (parity + time), asynchronous forward, bench.py with audited parity + slam block
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "quadrant_queue_backward or runs_ahead or capacity_overflow" > $O/pytest_d1.log 2>&1; tail -5 $O/pytest_d1.log | cut -c1-600
for rep in 1 2; do
ABL_TAG=bwd2 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_d.txt
VTGS_BWD_IMPL=3 ABL_TAG=bwd3 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_d.txt
done
VTGS_FORWARD_MODE=checked ABL_TAG=checked python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_d.txt
python tools/host_overhead.py 2>&1 | grep "host floor" | tee $O/host_d.txt
VTGS_FORWARD_MODE=checked python tools/host_overhead.py 2>&1 | grep "host floor" | tee -a $O/host_d.txt
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $O/pytest_d_full.log 2>&1; tail -5 $O/pytest_d_full.log | cut -c1-600
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench_d.json 2> $O/bench_d.err; tail -3 $O/bench_d.err; cut -c1-1500 $O/bench_d.json
