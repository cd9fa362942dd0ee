#!/bin/bash
R=$PWD; O=$R/gpurun_out/${VTGS_ROUND:-r5}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in ${MODES:-base fwdonly}; do
  rm -rf $O/bub_$m
  MODE=$m rocprofv3 --kernel-trace --output-format csv -d $O/bub_$m -o run -- python3 $R/tools/bubble_study.py > $O/bub_$m.log 2>&1 || { tail -5 $O/bub_$m.log; exit 1; }
  grep "step " $O/bub_$m.log
  f=$(find $O/bub_$m -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_gaps.py $f project_and_bin | grep -E "steps;|bubble" 
  rm -rf $O/bub_$m
done
for m in ${MODES:-base fwdonly}; do MODE=$m python3 $R/tools/bubble_study.py 2>&1 | grep "step "; done
