#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --slam-frames 2 > $O/bench_ap1.json 2> $O/bench_ap1.err || { tail -5 $O/bench_ap1.err; echo FAILED n1; exit 1; }
timeout -k 10 600 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --slam-frames 1 > $O/bench_ap2.json 2> $O/bench_ap2.err || { tail -15 $O/bench_ap2.err | cut -c1-300; echo FAILED n2; exit 1; }
python - <<'PY'
import json
for f in ("gpurun_out/r3/bench_ap1.json","gpurun_out/r3/bench_ap2.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(d["n_gpus"], d["ms_per_step"], d["slam"]["value"] if d["slam"] and "value" in d["slam"] else d["slam"], d.get("band",{}).get("replicated_frac"))
PY
