#!/bin/bash
# rocprofv3 passes of the bench command (run on the GPU box via gpurun).  Kernel trace + stats first,
# then separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md PMC slots).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
ARGS="--steps 20 --warmup 3 --no-cpu-baseline $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err || true
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err || true
find $OUT -name "*.csv" | head -20
