#!/bin/bash
# rocprofv3 kernel trace of bench.py's timed loop (no audit, no CPU baseline, no SLAM block) + the bubble summary of
# tools/trace_gaps.py over the FIRST <steps + warmup> steps (what follows in the trace is bench.py's per-kernel event phase)
#   (on the GPU box)  bash tools/trace_loop.sh <tag>
TAG=${1:-x}; R=$PWD; O=$R/gpurun_out/${VTGS_ROUND:-r5}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/trace_$TAG
rocprofv3 --kernel-trace --output-format csv -d $O/trace_$TAG -o run -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --audit-rows '' --slam-frames 0 > $O/trace_$TAG.log 2>&1 || { tail -5 $O/trace_$TAG.log; exit 1; }
grep -o '"ms_per_step": [0-9.]*' $O/trace_$TAG.log
cd $R; f=$(find $O/trace_$TAG -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f project_and_bin 60 | tee $O/trace_${TAG}_gaps.txt
