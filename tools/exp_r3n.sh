#!/bin/bash
# batch N: planned bins; workgroup-level early exit of the projection on a band; anisotropic render_frame
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_planned_bins.py tests/test_gpu_fused_frame.py tests/test_gpu_parity.py tests/test_gpu_abi_modes.py -q -m gpu > $O/pytest_n1.log 2>&1 || { tail -60 $O/pytest_n1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_n1.log
: > $O/bands_n.jsonl
for b in 0/8 3/8 7/8 2/4; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 10 --band $b >> $O/bands_n.jsonl 2>> $O/bench_n.err || { tail -5 $O/bench_n.err; fail "band $b"; }
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --band 3/8 --n 5000000 --width 1752 --height 1168 >> $O/bands_n.jsonl 2>> $O/bench_n.err || { tail -5 $O/bench_n.err; fail "band 5M"; }
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --slam-frames 0 --audit-rows '' --no-cpu-baseline >> $O/bands_n.jsonl 2>> $O/bench_n.err || { tail -5 $O/bench_n.err; fail "bench full"; }
VTGS_BINS=planned timeout -k 10 300 python bench.py --steps 30 --warmup 10 --slam-frames 0 --audit-rows '' --no-cpu-baseline >> $O/bands_n.jsonl 2>> $O/bench_n.err || { tail -5 $O/bench_n.err; fail "bench planned"; }
python - <<'PY'
import json
for ln in open("gpurun_out/r3/bands_n.jsonl"):
    if not ln.startswith("{"): continue
    d=json.loads(ln)
    print(d["config"]["gaussians"], d["config"]["mode"], d.get("band"), "ms/step", d["ms_per_step"], d["kernels_us"])
PY
