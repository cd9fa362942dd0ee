#!/bin/bash
# batch O: projection with the kernel-uniform band path, finalize with the coalesced plan rebuild
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_planned_bins.py tests/test_gpu_parity.py -q -m gpu -k "planned or band or binning or run_ahead or growing" > $O/pytest_o1.log 2>&1 || { tail -40 $O/pytest_o1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_o1.log
: > $O/bands_o.jsonl
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --slam-frames 0 --audit-rows '' --no-cpu-baseline >> $O/bands_o.jsonl 2>> $O/bench_o.err || { tail -5 $O/bench_o.err; fail "bench full"; }
for b in 0/8 3/8 7/8 2/4 1/2; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 10 --band $b >> $O/bands_o.jsonl 2>> $O/bench_o.err || { tail -5 $O/bench_o.err; fail "band $b"; }
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --band 3/8 --n 5000000 --width 1752 --height 1168 >> $O/bands_o.jsonl 2>> $O/bench_o.err || { tail -5 $O/bench_o.err; fail "band 5M"; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --n 5000000 --width 1752 --height 1168 --slam-frames 0 --audit-rows '' --no-cpu-baseline >> $O/bands_o.jsonl 2>> $O/bench_o.err || { tail -5 $O/bench_o.err; fail "5M full"; }
VTGS_BINS=planned timeout -k 10 300 python bench.py --steps 30 --warmup 10 --slam-frames 0 --audit-rows '' --no-cpu-baseline >> $O/bands_o.jsonl 2>> $O/bench_o.err || { tail -5 $O/bench_o.err; fail "bench planned"; }
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --slam-frames 0 --audit-rows '' --no-cpu-baseline >> $O/bands_o.jsonl 2>> $O/bench_o.err || { tail -5 $O/bench_o.err; fail "bench full 2"; }
python - <<'PY'
import json
for ln in open("gpurun_out/r3/bands_o.jsonl"):
    if not ln.startswith("{"): continue
    d=json.loads(ln)
    print(d["config"]["gaussians"], d["config"]["mode"], (d.get("band") or {}).get("tile_rows"), "ms/step", d["ms_per_step"], {k: round(v,1) for k,v in d["kernels_us"].items()})
PY
