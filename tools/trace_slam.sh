#!/bin/bash
# rocprofv3 kernel trace of two frames of the SLAM loop (bench_slam.py through the get_loss mirror) + per-iteration period,
# busy time, bubbles and kernel table (tools/trace_gaps.py keyed on the first kernel of an iteration)
#   (on the GPU box)  bash tools/trace_slam.sh <tag>
TAG=${1:-x}; R=$PWD; O=$R/gpurun_out/${VTGS_ROUND:-r5}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/slamtrace_$TAG
rocprofv3 --kernel-trace --output-format csv -d $O/slamtrace_$TAG -o run -- python3 $R/bench_slam.py --frames 2 --warmup-frames 1 --get-loss --shared-geometry --global-submaps 2 --base-frame-every 40 --emulate-window 12 > $O/slamtrace_$TAG.log 2>&1 || { tail -5 $O/slamtrace_$TAG.log; exit 1; }
cd $R; f=$(find $O/slamtrace_$TAG -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py $f project_and_bin | tee $O/slamtrace_${TAG}_gaps.txt
python - "$f" <<'PY' | tee -a $O/slamtrace_${TAG}_gaps.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'project_and_bin' in r['Kernel_Name']]
for which, a in (('an early iteration', idx[len(idx) // 4]), ('a late iteration', idx[-40])):
    b = idx[idx.index(a) + 1]
    print(f'--- {which}: kernels in order (start offset us, duration us, gap before us)')
    t0 = int(rows[a]['Start_Timestamp']); prev = None
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"  {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {0 if prev is None else (s - prev) / 1e3:6.1f}  {r['Kernel_Name'].split('(')[0][-70:]}")
        prev = e
PY
