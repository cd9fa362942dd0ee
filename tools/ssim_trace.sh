#!/bin/bash
# average duration of the two SSIM kernels at 1200x680 (rocprofv3 --kernel-trace --stats over tools/ssim_loop.py)
TAG=${1:-x}; R=$PWD; O=$R/gpurun_out/${VTGS_ROUND:-r5}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/ssim_trace_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ssim_trace_$TAG -o run -- python3 $R/tools/ssim_loop.py 60 > $O/ssim_trace_$TAG.log 2>&1 || { tail -5 $O/ssim_trace_$TAG.log; exit 1; }
cd $R; f=$(find $O/ssim_trace_$TAG -name "*kernel_stats.csv" | head -1); grep -i "ssim\|Name" $f | cut -d, -f1-6 | cut -c1-200
tail -1 $O/ssim_trace_$TAG.log
