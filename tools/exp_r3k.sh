#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "growing_scene or runs_ahead or replayed" > $O/pytest_k1.log 2>&1 || { tail -25 $O/pytest_k1.log | cut -c1-300; fail "soak test"; }
tail -2 $O/pytest_k1.log
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_slam_r3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_slam_r3 -o run -- python3 $R/bench_slam.py --frames 2 --get-loss > $O/prof_slam_r3.log 2>&1 || { tail -5 $O/prof_slam_r3.log; fail "rocprof slam"; }
cd $R
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3/prof_slam_r3/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
