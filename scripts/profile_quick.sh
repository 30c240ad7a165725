#!/bin/bash
# kernel stats of a short default bench run (no PMC)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_quick
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/run.json 2> $OUT/err.log || true
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:52].ljust(52), r["Calls"].rjust(6), "%10.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
PY
