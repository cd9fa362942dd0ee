#!/bin/bash
# One parametrised experiment runner for the GPU box (replaces the per-experiment tools/exp_r*.sh scripts of rounds 2-3).
#
#   tools/gpu_exp.sh <tag> <step> [<step> ...]        (run on the GPU box: gpurun -- 'bash tools/gpu_exp.sh r4a tests bench')
#
# Each step writes gpurun_out/<round>/<step>_<tag>.log; the script stops at the first failing step (no GPU step is started
# after a failed or timed-out one) and exits NON-ZERO when any log mentions a GPU memory access fault, an abort or a
# segmentation fault -- an experiment script of round 3 exited 0 through such a fault.
#
# steps:   tests            the whole -m gpu suite, one process
#          tests:<expr>     pytest -k <expr> of the -m gpu suite (commas for spaces: tests:fused_frame,or,poison)
#          smoke            __graft_entry__.smoke()
#          bench            python bench.py (default arguments)
#          bench:<args>     python bench.py <args>   (commas for spaces: bench:--steps,50,--mode,tracking)
#          timing           tools/kernel_timing.py at the headline shape (env ABL_* passes through)
#          timing:<lib>     the same with VTGS_LIBRARY=vtgaussian-slam_amd/lib/<lib>   (A/B builds on the same box)
#          stamps:<lib>     tools/forward_stamps.py with a -DVTGS_Q_STAMPS build
#          slam:<args>      python bench_slam.py <args>
#          py:<file>[:args] python <file> <args>
#          sh:<script>[:args] bash <script> <args>     (commas for spaces: sh:tools/trace_slam_late.sh:a,22,--densify)
#          profile          tools/profile_round.sh <tag>
set -o pipefail
TAG=$1; shift
ROUND=${VTGS_ROUND:-r6}; export VTGS_ROUND=$ROUND
R=$PWD; O=$R/gpurun_out/$ROUND; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
LIMIT=${VTGS_STEP_TIMEOUT:-900}
fail=0
run() {   # run <log> <command...>
  local log=$1; shift
  echo "== $* > $log"
  timeout -k 10 $LIMIT "$@" > $log 2>&1
  local rc=$?
  # a failing run keeps its log under a name no later run writes to (round 4: the fault text of DESIGN 7.6 was overwritten by
  # the passing re-run of the same step and tag)
  if [ $rc -ne 0 ] || grep -q -E "Memory access fault|Fatal Python error|Segmentation fault|core dumped|HSA_STATUS_ERROR" $log; then
    cp $log ${log%.log}.FAILED.$(date +%H%M%S).log
  fi
  if grep -q -E "Memory access fault|Fatal Python error|Segmentation fault|core dumped|HSA_STATUS_ERROR" $log; then
    echo "GPU FAULT / ABORT in $log:"; grep -E "Memory access fault|Fatal Python error|Segmentation fault|core dumped|HSA_STATUS_ERROR" $log | head -5
    # the runtime leaves a GPU core file (gpucore.<pid>) in the working directory: the waves in flight, their kernel and their
    # program counters name the faulting kernel -- the fault line itself only has an address (round 6)
    for c in $(ls -t gpucore.* 2>/dev/null | head -1); do
      echo "== rocgdb on $c (summary in ${log%.log}.gpucore.txt)"
      timeout -k 5 240 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "info agents" -ex "info threads" -ex "thread apply all bt 3" \
        $(command -v python3) $c 2>&1 | head -c 4000000 > ${log%.log}.gpucore.txt
      grep -E "fault|SIGSEGV|SIGBUS|exception|stopped|Memory" ${log%.log}.gpucore.txt | sort | uniq -c | sort -rn | head -8
      grep -o -E "in [A-Za-z_0-9:<>, ]+\(" ${log%.log}.gpucore.txt | sort | uniq -c | sort -rn | head -8
      rm -f $c
    done
    fail=2; return 2
  fi
  if [ $rc -ne 0 ]; then echo "step failed (rc $rc): $*"; tail -25 $log | cut -c1-300; fail=1; return 1; fi
  tail -${VTGS_TAIL:-3} $log | cut -c1-1500
  return 0
}
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  case $name in
    tests)
             # --capture=sys: pytest's default capture redirects FILE DESCRIPTOR 2 into a temporary file for the duration of a test and
             # shows it only for a failed one -- when the process ABORTS the file is lost with it, which is why the two aborts above left
             # nothing but faulthandler's dump (it writes to a duplicate of the original descriptor): whatever the runtime printed
             # (a memory-fault line, a queue error) went into the capture.  Capturing sys.stdout / sys.stderr only lets it through.
             if [ -n "$arg" ]; then run $O/tests_$TAG.log python -m pytest tests/ -x -q -m gpu --capture=sys -k "${arg//,/ }"; else run $O/tests_$TAG.log python -m pytest tests/ -x -q -m gpu --capture=sys; fi ;;
    smoke)   run $O/smoke_$TAG.log python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench)   run $O/bench_$TAG.log python bench.py ${arg//,/ } ;;
    timing)  if [ -n "$arg" ]; then VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/$arg ABL_TAG=$arg run $O/timing_${TAG}_${arg%.so}.log python tools/kernel_timing.py; else ABL_TAG=shipped run $O/timing_$TAG.log python tools/kernel_timing.py; fi ;;
    stamps)  VTGS_LIBRARY=$R/vtgaussian-slam_amd/lib/$arg VTGS_TAIL=40 run $O/stamps_${TAG}.log python tools/forward_stamps.py ;;
    slam)    run $O/slam_$TAG.log python bench_slam.py ${arg//,/ } ;;
    py)      f=${arg%%:*}; a=""; [ "$arg" != "$f" ] && a=${arg#*:}; VTGS_TAIL=${VTGS_TAIL:-30} run $O/py_${TAG}_$(basename $f .py).log python $f ${a//,/ } ;;
    sh)      f=${arg%%:*}; a=""; [ "$arg" != "$f" ] && a=${arg#*:}; VTGS_TAIL=${VTGS_TAIL:-60} run $O/sh_${TAG}_$(basename $f .sh).log bash $f ${a//,/ } ;;
    profile) run $O/profile_$TAG.log bash tools/profile_round.sh $TAG ;;
    *) echo "unknown step $step"; exit 64 ;;
  esac
  [ $fail -ne 0 ] && break
done
exit $fail
