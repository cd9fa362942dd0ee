#!/bin/bash
# rocprofv3 kernel statistics of the fused SLAM loop (bench_slam.py through the get_loss mirror): which kernels an iteration is made of.
#   bash tools/profile_slam.sh <tag> [bench_slam.py arguments]        -> gpurun_out/prof_slam_<tag>/run_kernel_stats.csv
set -o pipefail
TAG=${1:-r4}; shift
R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_slam_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slam_$TAG -o run -- python3 $R/bench_slam.py --frames 2 --warmup-frames 1 --get-loss "$@" > $O/prof_slam_$TAG.log 2>&1 || { echo "profile failed"; tail -5 $O/prof_slam_$TAG.log; exit 1; }
cd $R
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/prof_slam_$TAG/run_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel | calls | avg us | total ms | %")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f'{r["Name"][:70]:70s} | {r["Calls"]:>6s} | {float(r["AverageNs"]) / 1e3:8.1f} | {float(r["TotalDurationNs"]) / 1e6:8.2f} | {100 * float(r["TotalDurationNs"]) / tot:5.1f}')
PY
tail -1 $O/prof_slam_$TAG.log | cut -c1-300
