#!/bin/bash
# rocprofv3 kernel statistics of (1) the linear assembly at 3.3 M elements, (2) two Newton iterations of the nonlinear loop,
# (3) the NDOF = 1 and NDOF = 6 generic-block solves (gpurun).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_asm
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/asm -- python3 $R/scripts/bench_assembly.py 149 > $OUT/asm.log 2> $OUT/asm.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nl -- python3 $R/scripts/bench_nonlinear.py 149 1 1 2 > $OUT/nl.log 2> $OUT/nl.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nn1 -- python3 $R/scripts/bench_nn.py 1 150 > $OUT/nn1.log 2> $OUT/nn1.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nn6 -- python3 $R/scripts/bench_nn.py 6 70 > $OUT/nn6.log 2> $OUT/nn6.err || true
for d in asm nl nn1 nn6; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; done
ls -la $OUT/*.csv
