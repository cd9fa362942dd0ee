#!/bin/bash
# Measured set behind profiles/<tag>_*: smoke, bench line, rocprofv3 kernel stats, HBM traffic (two PMC passes) and SQ
# counters (two PMC passes) of the same command.  Run on the GPU box from the repository root:
#     bash tools/profile_round.sh r2a        then, back in the build container:  python tools/summarize_profiles.py r2a
set -o pipefail
TAG=${1:-r2x}
R=$PWD; O=$R/gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $O/smoke_$TAG.txt || exit 1
python bench.py 2> $O/bench_$TAG.err | tee $O/bench_$TAG.json | cut -c1-400 || exit 1
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_$TAG $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -o run -- python3 $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --audit-rows '' --slam-frames 0 > $O/prof_$TAG.log 2>&1 || { echo "kernel-trace pass failed"; tail -5 $O/prof_$TAG.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o run -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --audit-rows '' --slam-frames 0 > $O/pmc_fetch.log 2>&1 || { echo "FETCH_SIZE pass failed"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o run -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --audit-rows '' --slam-frames 0 > $O/pmc_write.log 2>&1 || { echo "WRITE_SIZE pass failed"; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_sq1 -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --audit-rows '' --slam-frames 0 > $O/pmc_sq1.log 2>&1 || echo "SQ pass 1 failed"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq2 -o run -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --audit-rows '' --slam-frames 0 > $O/pmc_sq2.log 2>&1 || echo "SQ pass 2 failed"
cd $R; ls $O/prof_$TAG $O/pmc_fetch $O/pmc_write | head -20
