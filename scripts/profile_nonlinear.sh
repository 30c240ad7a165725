#!/bin/bash
# rocprofv3 kernel statistics of two Newton iterations of the nonlinear static loop at 10.1M DOF (gpurun).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_nl
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/bench_nonlinear.py 149 1 1 2 > $OUT/run.log 2> $OUT/trace.err || true
find $OUT -name "*kernel_stats.csv"
