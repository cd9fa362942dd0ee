#!/bin/bash
# experiment batch r2a (run on the GPU box through gpurun): quadrant-queue forward, occupancy variants, SQ counters
set -o pipefail
R=$PWD; O=$R/gpurun_out/r2; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "quadrant" > $O/pytest_q2.log 2>&1; tail -3 $O/pytest_q2.log | cut -c1-300
for lib in libvtgs libvtgs_q3 libvtgs_q5 libvtgs_q6; do
  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/$lib.so VTGS_FWD_IMPL=3 ABL_TAG=$lib-fwd3 python tools/kernel_timing.py 2>&1 | grep step
done | tee $O/timing_q2.txt
VTGS_FWD_IMPL=2 ABL_TAG=fwd2 python tools/kernel_timing.py 2>&1 | grep step | tee -a $O/timing_q2.txt
cd /tmp; export TMPDIR=/tmp
for pass in a b; do
  if [ $pass = a ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA";
  else C="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; fi
  for impl in 3 2; do
    VTGS_FWD_IMPL=$impl ABL_BWD=0 rocprofv3 --pmc $C -d $O/sq_${pass}_fwd$impl -o run --output-format csv -- python3 $R/tools/kernel_timing.py > $O/sq_${pass}_fwd$impl.log 2>&1 || echo "pmc pass $pass impl $impl failed"
  done
done
cd $R
python tools/sq_counters.py $O/sq_a_fwd3 $O/sq_b_fwd3 > $O/sq_fwd3.md; python tools/sq_counters.py $O/sq_a_fwd2 $O/sq_b_fwd2 > $O/sq_fwd2.md
cat $O/sq_fwd3.md $O/sq_fwd2.md | cut -c1-400
