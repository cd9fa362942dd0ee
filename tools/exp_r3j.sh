#!/bin/bash
# r3j: forward prologue with three chunks' records requested together; A/B against the previous commit's library
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
L=$R/vtgaussian-slam_amd/lib
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "parity or cfg_a or odd_shapes or sort or quadrant or giant or saturating" > $O/pytest_j1.log 2>&1 || { tail -15 $O/pytest_j1.log | cut -c1-300; fail "parity subset"; }
tail -2 $O/pytest_j1.log
for rep in 1 2; do
ABL_TAG=new python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_j.txt
VTGS_LIBRARY=$L/libvtgs_prev.so ABL_TAG=prev python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_j.txt
done
VTGS_LIBRARY=$L/libvtgs_stamps.so python tools/forward_stamps.py 2>&1 | grep -v "^backward" | head -8 | tee $O/stamps_j.txt
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/pytest_j_full.log 2>&1 || { tail -15 $O/pytest_j_full.log | cut -c1-300; fail "full suite"; }
tail -2 $O/pytest_j_full.log
python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-cpu-baseline --audit-rows '' --slam-frames 0 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err || { tail -5 $O/bench_2rank_gloo.err; fail "2-rank gloo"; }
cut -c1-400 $O/bench_2rank_gloo.json
