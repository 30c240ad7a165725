#!/bin/bash
# Round-2 measurement batch B (GPU box, via gpurun): rocprofv3 kernel statistics and PMC passes of the final build.
# Kernel trace + stats and the two PMC passes are separate runs (MI355X_MICROARCH.md PMC slots; gpurun refuses mixed modes).
# Outputs under gpurun_out/r02prof/; scripts/summarize_profiles.py r02 copies the summaries into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err && echo trace ok && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err && echo fetch ok && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err && echo write ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ilu -- python3 $R/bench.py --method 2 --precond 10 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/ilu.json 2> $OUT/ilu.err && echo ilu ok && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/ilu_fetch -- python3 $R/bench.py --method 2 --precond 10 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/ilu_fetch.json 2> $OUT/ilu_fetch.err && echo ilu fetch ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nl -- python3 $R/scripts/bench_nonlinear.py 149 2 10 2 > $OUT/nl.log 2> $OUT/nl.err && echo nl ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/asm -- python3 $R/scripts/bench_assembly.py 149 > $OUT/asm.log 2> $OUT/asm.err && echo asm ok && \
FX_EISENSTAT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eis -- python3 $R/bench.py $ARGS > $OUT/eis.json 2> $OUT/eis.err && echo eis ok && \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/nn1 -- python3 $R/scripts/bench_nn.py 1 150 > $OUT/nn1.log 2> $OUT/nn1.err && echo nn1 ok
for d in ilu nl asm nn1 eis; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; done
f=$(find $OUT/ilu_fetch -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 - "$f" <<'PY' > $OUT/ilu_fetch_summary.txt
import collections, csv, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
print("FETCH_SIZE per launch (KiB, x2 corrected -> MB), BiCGSTAB + ILU(0) 10.1M DOF")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:10]:
    m = sum(v) / len(v)
    print("  %-60s launches=%4d  %12.1f KiB -> %9.1f MB" % (k[:60], len(v), m, 2 * m * 1024 / 1e6))
PY
ls $OUT/*.csv $OUT/*.txt 2>/dev/null
