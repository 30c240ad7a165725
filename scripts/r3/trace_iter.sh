#!/bin/bash
# Kernel timeline of the timed loop (rocprofv3 --kernel-trace): per-kernel durations AND the gaps between consecutive
# dispatches, which --stats does not show.  usage: trace_iter.sh <tag> <bench args...>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
OUT=$R/gpurun_out/r3/trace_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.txt || { tail -20 $OUT/err.txt; exit 1; }
F=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 $R/scripts/r3/trace_gaps.py $F > $OUT/gaps.txt
rm -rf $OUT/t
tail -80 $OUT/gaps.txt
