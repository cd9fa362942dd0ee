set -o pipefail
export VTGS_SLAM_VERBOSE=1
O=gpurun_out/r4; mkdir -p $O
python bench_slam.py --frames 3 --get-loss > $O/slam_a.json 2> $O/slam_a.err || exit 1
python bench_slam.py --frames 3 --get-loss --base-frame-every 3 --emulate-window 12 > $O/slam_b.json 2> $O/slam_b.err || exit 1
python bench_slam.py --frames 3 --get-loss --base-frame-every 3 --emulate-window 12 --global-submaps 2 > $O/slam_c.json 2> $O/slam_c.err || exit 1
VTGS_FORWARD_MODE=checked python bench_slam.py --frames 3 --get-loss > $O/slam_d.json 2> $O/slam_d.err || exit 1
for f in a b c d; do grep "bench_slam" $O/slam_$f.err | cut -c1-200; python -c "
import json,sys; d=json.load(open('$O/slam_$f.json')); print('$f', d['value'], d['tracking_ms_per_iter'], d['mapping_ms_per_iter'], d.get('regimes'))"; done
