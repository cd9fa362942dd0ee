#!/bin/bash
# batch M: y-cull of project_and_bin on a rank of the partition; band tests; per-band kernel shares again
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_frame.py tests/test_band_loss_gpu.py tests/test_gpu_configs.py -q -m gpu > $O/pytest_m1.log 2>&1 || { tail -40 $O/pytest_m1.log | cut -c1-300; fail "tests"; }
tail -2 $O/pytest_m1.log
: > $O/bands_m.jsonl
for b in 0/8 3/8 7/8 1/2 2/4; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 10 --band $b >> $O/bands_m.jsonl 2>> $O/bench_m.err || { tail -5 $O/bench_m.err; fail "band $b"; }
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --band 3/8 --n 5000000 --width 1752 --height 1168 >> $O/bands_m.jsonl 2>> $O/bench_m.err || { tail -5 $O/bench_m.err; fail "band 5M"; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --band 3/8 --mode mapping >> $O/bands_m.jsonl 2>> $O/bench_m.err || { tail -5 $O/bench_m.err; fail "band mapping"; }
python - <<'PY'
import json
for ln in open("gpurun_out/r3/bands_m.jsonl"):
    if not ln.startswith("{"): continue
    d=json.loads(ln)
    print(d["config"]["gaussians"], d["config"]["mode"], d["band"], "ms/step", d["ms_per_step"], d["kernels_us"])
PY
