#!/bin/bash
# Round-4 measurement batch (GPU box, via gpurun): bench lines + rocprofv3 kernel statistics + PMC passes of the FINAL build.
# Kernel trace + stats and each PMC counter are separate runs (MI355X_MICROARCH.md PMC slots; gpurun refuses mixed modes).
# Outputs under gpurun_out/r4prof/; scripts/r4/summarize.py copies the summaries into profiles/r04_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4prof
rm -rf $OUT; mkdir -p $OUT
cd $R
step() { echo "== $1"; }
step "bench lines"
timeout -k 10 300 python3 bench.py > $OUT/bench_10m_cg_ssor.json 2> $OUT/bench_10m_cg_ssor.err && \
timeout -k 10 300 python3 bench.py --standard --no-cpu-baseline > $OUT/bench_10m_cg_ssor_standard.json 2> $OUT/bstd.err && \
timeout -k 10 300 python3 bench.py --elems 69 --precond 3 > $OUT/bench_1m_cg_diag.json 2> $OUT/b1m.err && \
timeout -k 10 300 python3 bench.py --method 2 --precond 10 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_10m_bicgstab_ilu0.json 2> $OUT/bilu.err && \
FX_SSOR_NATURAL=1 timeout -k 10 300 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_10m_cg_ssor_natural.json 2> $OUT/bnat.err && \
timeout -k 10 300 python3 scripts/bench_nonlinear.py 149 1 1 3 2>/dev/null | tail -1 > $OUT/nonlinear_newton_10m.json && \
timeout -k 10 300 python3 scripts/bench_nonlinear.py 149 2 10 3 2>/dev/null | tail -1 > $OUT/nonlinear_newton_10m_bicgstab_ilu0.json && \
timeout -k 10 200 python3 scripts/bench_assembly.py 149 > $OUT/assembly_10m.json 2>/dev/null && \
timeout -k 10 200 python3 scripts/r4/bench_update_linear.py 149 > $OUT/update_linear_10m.json 2>/dev/null && echo "bench ok"
cd /tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline"
step "rocprof"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err && echo trace ok && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err && echo fetch ok && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err && echo write ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ilu -- python3 $R/bench.py --method 2 --precond 10 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/ilu.json 2> $OUT/ilu.err && echo ilu ok && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ilu_fetch -- python3 $R/bench.py --method 2 --precond 10 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/ilu_fetch.json 2> $OUT/ilu_fetch.err && echo ilu fetch ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/asm -- python3 $R/scripts/bench_assembly.py 149 > $OUT/asm.log 2> $OUT/asm.err && echo asm ok
for d in trace ilu asm; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; done
for d in pmc_fetch pmc_write ilu_fetch; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_counters.csv; done
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/ilu $OUT/ilu_fetch $OUT/asm
ls -la $OUT
