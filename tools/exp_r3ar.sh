#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > $O/pytest_ar_full.log 2>&1 || { tail -30 $O/pytest_ar_full.log | cut -c1-300; echo FAILED; exit 1; }
tail -2 $O/pytest_ar_full.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
