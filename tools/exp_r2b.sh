#!/bin/bash
# experiment batch r2b: lean quadrant-queue forward (64 VGPRs, 6 waves/SIMD)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "quadrant or variants_agree" > $O/pytest_q3.log 2>&1; tail -5 $O/pytest_q3.log | cut -c1-300
VTGS_FWD_IMPL=3 ABL_TAG=fwd3 python tools/kernel_timing.py 2>&1 | grep step | tee $O/timing_q3.txt
VTGS_FWD_IMPL=2 ABL_TAG=fwd2 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_q3.txt
VTGS_FWD_IMPL=3 ABL_TAG=fwd3-sat ABL_OPACITY=0.9 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_q3.txt
VTGS_FWD_IMPL=2 ABL_TAG=fwd2-sat ABL_OPACITY=0.9 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_q3.txt
cd /tmp; export TMPDIR=/tmp
for pass in a b; do
  if [ $pass = a ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA";
  else C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; fi
  VTGS_FWD_IMPL=3 ABL_BWD=0 rocprofv3 --pmc $C -d $O/sq3_${pass} -o run --output-format csv -- python3 $R/tools/kernel_timing.py > $O/sq3_${pass}.log 2>&1 || echo "pmc pass $pass failed"
done
cd $R
python tools/sq_counters.py $O/sq3_a $O/sq3_b > $O/sq3.md; grep "composite\|kernel" $O/sq3.md | cut -c1-330
cd tests/micro && hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result step_rate.hip -o /tmp/step_rate.bin 2>/dev/null && timeout -k 5 120 /tmp/step_rate.bin | tee $O/step_rate2.txt | grep "mode 0"
