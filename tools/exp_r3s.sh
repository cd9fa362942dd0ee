#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
: > $O/bench_s.txt
for a in "--steps 50 --warmup 10" "--steps 30 --warmup 10" "--steps 50 --warmup 10" "--steps 200 --warmup 50"; do
  timeout -k 10 200 python bench.py $a --slam-frames 0 --audit-rows '' --no-cpu-baseline 2>> $O/bench_s.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', d['ms_per_step'], d['kernels_us'])" >> $O/bench_s.txt || { tail -3 $O/bench_s.err; exit 1; }
done
VTGS_FORWARD_MODE=checked timeout -k 10 200 python bench.py --steps 50 --warmup 10 --slam-frames 0 --audit-rows '' --no-cpu-baseline 2>> $O/bench_s.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('checked', d['ms_per_step'], d['kernels_us'])" >> $O/bench_s.txt
timeout -k 10 200 python tools/host_overhead.py >> $O/bench_s.txt 2>&1
cat $O/bench_s.txt
