#!/bin/bash
# batch AH: final validation of the round's last build + the band lines for DESIGN 5
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_ah_full.log 2>&1 || { tail -40 $O/pytest_ah_full.log | cut -c1-300; fail "gpu suite"; }
tail -2 $O/pytest_ah_full.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || fail smoke
: > $O/bands_ah.jsonl
for b in 0/8 3/8 7/8 2/4 1/2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 10 --band $b >> $O/bands_ah.jsonl 2>> $O/bench_ah.err || { tail -5 $O/bench_ah.err; fail "band $b"; }
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --band 3/8 --n 5000000 --width 1752 --height 1168 >> $O/bands_ah.jsonl 2>> $O/bench_ah.err || { tail -5 $O/bench_ah.err; fail "band 5M"; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --band 1/4 --n 2000000 --width 640 --height 480 >> $O/bands_ah.jsonl 2>> $O/bench_ah.err || { tail -5 $O/bench_ah.err; fail "band 2M"; }
python - <<'PY'
import json
for ln in open("gpurun_out/r3/bands_ah.jsonl"):
    if not ln.startswith("{"): continue
    d=json.loads(ln)
    print(d["config"]["gaussians"], d["config"]["mode"], d["band"]["tile_rows"], "ms/step", d["ms_per_step"], "replicated", d["band"]["replicated_frac"], {k: round(v,1) for k,v in d["kernels_us"].items()})
PY
timeout -k 10 900 python bench.py > $O/bench_ah.json 2>> $O/bench_ah.err || { tail -5 $O/bench_ah.err; fail bench; }
tail -1 $O/bench_ah.json | cut -c1-300
