#!/bin/bash
# rocprofv3 kernel statistics of BiCGSTAB + ILU(0) iterations at 10.1M DOF (level-scheduled sweeps).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_ilu
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --method 2 --precond 10 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/run.log 2> $OUT/trace.err || true
find $OUT -name "*kernel_stats.csv"
