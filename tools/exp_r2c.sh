#!/bin/bash
# experiment batch r2c: phased quadrant forward (6 and 5 waves), MFMA grouping in the lane = pixel kernels
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "quadrant or variants_agree" > $O/pytest_q4.log 2>&1; tail -4 $O/pytest_q4.log | cut -c1-300
VTGS_FWD_IMPL=3 ABL_TAG=q6 python tools/kernel_timing.py 2>&1 | grep step | tee $O/timing_q4.txt
VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_q5.so VTGS_FWD_IMPL=3 ABL_TAG=q5 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_q4.txt
VTGS_FWD_IMPL=2 ABL_TAG=px python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_q4.txt
VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_pxg.so VTGS_FWD_IMPL=2 ABL_TAG=px-grouped python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_q4.txt
VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/libvtgs_pxg.so python -m pytest tests/test_gpu_parity.py -q -m gpu -k "forward_backward_parity or saturating" > $O/pytest_pxg.log 2>&1; tail -3 $O/pytest_pxg.log | cut -c1-300
cd /tmp; export TMPDIR=/tmp
for pass in a b; do
  if [ $pass = a ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA";
  else C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; fi
  VTGS_FWD_IMPL=3 ABL_BWD=0 rocprofv3 --pmc $C -d $O/sq4_${pass} -o run --output-format csv -- python3 $R/tools/kernel_timing.py > $O/sq4_${pass}.log 2>&1 || echo "pmc pass $pass failed"
done
cd $R
python tools/sq_counters.py $O/sq4_a $O/sq4_b > $O/sq4.md; grep "composite" $O/sq4.md | cut -c1-330
