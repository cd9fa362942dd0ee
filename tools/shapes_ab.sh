#!/bin/bash
# 5 M @ 1752x1168 through tools/kernel_timing.py with the windowed LDS tile table (default) and with global-atomic bins
for r in 1 2; do
  for impl in 1 0; do
    echo "== VTGS_BIN_IMPL=$impl"; VTGS_BIN_IMPL=$impl ABL_N=5000000 ABL_W=1752 ABL_H=1168 ABL_TAG=scannetpp_bin$impl python tools/kernel_timing.py 2>&1 | tail -1
  done
done
