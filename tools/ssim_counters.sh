#!/bin/bash
# SQ counters of the two SSIM kernels (two rocprofv3 --pmc passes over tools/ssim_loop.py)
TAG=${1:-x}; R=$PWD; O=$R/gpurun_out/${VTGS_ROUND:-r5}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/ssim_sq1_$TAG $O/ssim_sq2_$TAG
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/ssim_sq1_$TAG -o run -- python3 $R/tools/ssim_loop.py 5 > $O/ssim_sq1_$TAG.log 2>&1 || { tail -5 $O/ssim_sq1_$TAG.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/ssim_sq2_$TAG -o run -- python3 $R/tools/ssim_loop.py 5 > $O/ssim_sq2_$TAG.log 2>&1 || { tail -5 $O/ssim_sq2_$TAG.log; exit 1; }
cd $R; python tools/sq_counters.py $O/ssim_sq1_$TAG $O/ssim_sq2_$TAG | tee $O/ssim_counters_$TAG.md
