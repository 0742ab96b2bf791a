#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag> [workload ...]     e.g. r03 c3 c2 c5fp8 c5bf16
#   -> gpurun_out/prof_<tag>/<workload>/{trace,FETCH_SIZE,WRITE_SIZE,MFMA}/...   (tools/profile_collect.py copies the summaries into profiles/)
# Four separate rocprofv3 runs of the SAME bench command per workload (the guide's HBM/rocprofv3 recipe: --pmc passes apart from the trace):
#   1. --kernel-trace --stats                     per-kernel durations inside real train steps
#   2. --pmc FETCH_SIZE                           HBM read side (x2 on gfx950)
#   3. --pmc WRITE_SIZE                           HBM write side
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16      matrix-core busy cycles and executed MFMA ops
# The program itself follows `--` (no wrapper: the profiler's library initialises the GPU before the program starts).
set -e
TAG=${1:-r03}
shift || true
WLS=${@:-c3}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for WL in $WLS; do
  OUT=$ROOT/gpurun_out/prof_$TAG/$WL
  mkdir -p $OUT
  case $WL in c3|c2) STEPS=60; WARM=10; PSTEPS=5;; *) STEPS=12; WARM=4; PSTEPS=3;; esac
  ARGS="--workload $WL --no-cpu-baseline --no-roofline --no-configs"
  unset EAE_FORK_EVENTS
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --steps $STEPS --warmup $WARM $ARGS > $OUT/trace.log 2>&1
  # Counter collection runs ONE kernel at a time on the device: a gate kernel (side stream polling the main stream's progress word)
  # would spin until its bounded time-out while the kernel it waits for cannot start.  The counter passes therefore use event-based
  # hand-overs (EAE_FORK_EVENTS=1: same kernels, same arguments, same per-kernel traffic; exported here, not through an `env` hop).
  export EAE_FORK_EVENTS=1
  for P in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/$P -o p -- python3 $ROOT/bench.py --steps $PSTEPS --warmup 2 $ARGS > $OUT/$P.log 2>&1
  done
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/MFMA -o p -- python3 $ROOT/bench.py --steps $PSTEPS --warmup 2 $ARGS > $OUT/MFMA.log 2>&1
  echo "$WL steps=$STEPS warmup=$WARM pmc_steps=$PSTEPS" > $OUT/run.txt
  tail -1 $OUT/trace.log | cut -c1-200
  # keep the merge-back small: the per-dispatch traces of the counter passes are only needed as counter_collection.csv
  find $OUT -name "*_kernel_trace.csv" -path "*SIZE*" -delete; find $OUT -name "*_kernel_trace.csv" -path "*MFMA*" -delete
done
find $ROOT/gpurun_out/prof_$TAG -name "*.csv" | head -40
