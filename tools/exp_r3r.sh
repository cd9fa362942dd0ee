#!/bin/bash
# batch R: full GPU suite + smoke + default bench on the current build
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
fail() { echo "FAILED: $1"; exit 1; }
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/pytest_r_full.log 2>&1 || { tail -40 $O/pytest_r_full.log | cut -c1-300; fail "gpu suite"; }
tail -2 $O/pytest_r_full.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || fail smoke
timeout -k 10 900 python bench.py > $O/bench_r.json 2> $O/bench_r.err || { tail -5 $O/bench_r.err; fail bench; }
tail -1 $O/bench_r.json | cut -c1-2500
