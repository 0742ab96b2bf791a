#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>      e.g. r02 -> gpurun_out/prof_<tag>/...
# Four separate rocprofv3 runs of the SAME bench command (the guide's HBM/rocprofv3 recipe: --pmc passes apart from the trace):
#   1. --kernel-trace --stats                     per-kernel durations inside real train steps
#   2. --pmc FETCH_SIZE                           HBM read side (x2 on gfx950)
#   3. --pmc WRITE_SIZE                           HBM write side
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16      matrix-core busy cycles and executed MFMA ops
# The program itself follows `--` (no wrapper: the profiler's library initialises the GPU before the program starts).
set -e
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-configs > $OUT/trace.log 2>&1
# Counter collection runs ONE kernel at a time on the device: a gate kernel (side stream polling the main stream's progress word)
# would spin until its bounded time-out while the kernel it waits for cannot start.  The counter passes therefore use event-based
# hand-overs (EAE_FORK_EVENTS=1: same kernels, same arguments, same per-kernel traffic; exported here, not through an `env` hop).
export EAE_FORK_EVENTS=1
python3 -c "import os; print({k: v for k, v in os.environ.items() if 'ROCP' in k.upper()})" > $OUT/env_plain.log 2>&1 || true
for P in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/$P -o p -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-configs > $OUT/$P.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/MFMA -o p -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-configs > $OUT/MFMA.log 2>&1
find $OUT -name "*.csv" | head -20
