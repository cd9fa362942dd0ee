#!/bin/bash
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --slam-frames 0 > $O/bench_ab.json 2> $O/bench_ab.err || { tail -5 $O/bench_ab.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/bench_ab.json").read().strip().splitlines()[-1])
p=d["parity"]
print({k:p[k] for k in ("grad_p999","grad_max_rel","grad_rel_l2","img_max_rel","unexplained","seconds")})
print(p["float32_oracle_vs_float64"])
PY
