#!/bin/bash
# batch L: band loss kernels, nullable backward outputs, N-rank rehearsal (gloo on one GPU), per-band kernel shares
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 420 python -m pytest tests/test_band_loss_gpu.py tests/test_get_loss_fixtures.py tests/test_gpu_fused_frame.py tests/test_gpu_losses.py tests/test_gpu_abi_modes.py -q -m gpu -x > $O/pytest_l1.log 2>&1 || { tail -40 $O/pytest_l1.log | cut -c1-300; fail "band / loss tests"; }
tail -2 $O/pytest_l1.log
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --slam-frames 0 --audit-rows '' --no-cpu-baseline > $O/bench_l_rasterize.json 2> $O/bench_l.err || { tail -5 $O/bench_l.err; fail "bench rasterize"; }
timeout -k 10 300 python bench.py --steps 30 --warmup 10 --mode tracking --slam-frames 0 --audit-rows '' --no-cpu-baseline > $O/bench_l_tracking.json 2>> $O/bench_l.err || { tail -5 $O/bench_l.err; fail "bench tracking"; }
python - <<'PY'
import json
for m in ("rasterize","tracking"):
    d=json.loads(open(f"gpurun_out/r3/bench_l_{m}.json").read().strip().splitlines()[-1])
    print(m, d["ms_per_step"], d["kernels_us"])
PY
: > $O/bands_l.jsonl
for b in 0/8 3/8 7/8 1/2 2/4; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 10 --band $b >> $O/bands_l.jsonl 2>> $O/bench_l.err || { tail -5 $O/bench_l.err; fail "band $b"; }
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --band 3/8 --n 5000000 --width 1752 --height 1168 >> $O/bands_l.jsonl 2>> $O/bench_l.err || { tail -5 $O/bench_l.err; fail "band 5M"; }
python - <<'PY'
import json
for ln in open("gpurun_out/r3/bands_l.jsonl"):
    if not ln.startswith("{"): continue
    d=json.loads(ln)
    print(d["config"]["gaussians"], d["band"], "ms/step", d["ms_per_step"], d["kernels_us"])
PY
# two ranks on this GPU: the N-rank code path of bench.py (timed region + slam block), collectives through the host
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --slam-frames 1 > $O/bench_l_gloo2.json 2> $O/bench_l_gloo2.err || { tail -15 $O/bench_l_gloo2.err | cut -c1-300; fail "gloo 2 ranks"; }
tail -1 $O/bench_l_gloo2.json | cut -c1-1500
