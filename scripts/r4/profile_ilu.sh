#!/bin/bash
# Round-4 measurement batch for the level-scheduled sweeps (GPU box, via gpurun), after the batched polling passes of k_tri_dataflow:
# bench lines (BiCGSTAB + ILU(0), CG + natural-order SSOR), rocprofv3 kernel statistics of the ILU(0) run, the nonlinear Newton loop with
# ILU(0), and the plane march (csrc/fx_march.h) with its per-chunk / per-round traces.  Outputs under gpurun_out/r4ilu/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4ilu
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/r4
cd $R
timeout -k 10 300 python3 bench.py --method 2 --precond 10 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_10m_bicgstab_ilu0.json 2> $OUT/bilu.err && echo ilu ok && \
FX_SSOR_NATURAL=1 timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_10m_cg_ssor_natural.json 2> $OUT/bnat.err && echo natural ok && \
timeout -k 10 300 python3 scripts/bench_nonlinear.py 149 2 10 3 2>/dev/null | tail -1 > $OUT/nonlinear_newton_10m_bicgstab_ilu0.json && echo newton ok && \
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ilu -- python3 $R/bench.py --method 2 --precond 10 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/ilu.json 2> $OUT/ilu.err && echo trace ok && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ilu_fetch -- python3 $R/bench.py --method 2 --precond 10 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/ilu_fetch.json 2> $OUT/ilu_fetch.err && echo fetch ok && \
cd $R && MARCH_TRACE=1 MARCH_ROUNDS=0,6,60,240,450 timeout -k 10 300 python3 scripts/r4/march_bench.py 3750:2 4500:2 22500:3 0:0 > $OUT/march_ilu0.txt 2>&1 && echo march ok && \
MARCH_TRACE=1 timeout -k 10 200 python3 scripts/r4/march_bench.py --precond 1 3750:2 > $OUT/march_ssor_natural.txt 2>&1 && echo march natural ok
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/ilu_kernel_stats.csv \; 2>/dev/null
find $OUT/ilu_fetch -name "*counter_collection.csv" -exec cp {} $OUT/ilu_fetch_counters.csv \; 2>/dev/null
ls $OUT
