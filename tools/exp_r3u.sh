#!/bin/bash
# batch U: rehearsal of the N-rank bench path on one GPU (gloo), 2 and 4 ranks, final build
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --slam-frames 1 > $O/bench_u_gloo2.json 2> $O/bench_u_gloo2.err || { tail -15 $O/bench_u_gloo2.err | cut -c1-300; fail "gloo 2 ranks (self-launched)"; }
tail -1 $O/bench_u_gloo2.json | cut -c1-600
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --backend gloo --steps 10 --warmup 3 --slam-frames 1 --mode mapping > $O/bench_u_gloo4.json 2> $O/bench_u_gloo4.err || { tail -15 $O/bench_u_gloo4.err | cut -c1-300; fail "gloo 4 ranks"; }
tail -1 $O/bench_u_gloo4.json | cut -c1-600
python - <<'PY'
import json
for f in ("gpurun_out/r3/bench_u_gloo2.json","gpurun_out/r3/bench_u_gloo4.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(d["n_gpus"], d["config"]["mode"], d["ms_per_step"], d["band"], d["slam"]["value"], d["slam"]["pose_error_after_tracking_cm_deg"])
PY
