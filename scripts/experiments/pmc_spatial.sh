#!/bin/bash
# FETCH_SIZE of the SpMV with ascending vs spatial slice order (FX_SPMV_SPATIAL), 10.1M DOF.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  export FX_SPMV_SPATIAL=$v
  OUT=$R/gpurun_out/pmc_spatial_$v
  mkdir -p $OUT
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/run.json 2> $OUT/err.log || true
done
python3 - <<PY
import csv, glob, collections
for v in (0, 1):
    f = glob.glob("$R/gpurun_out/pmc_spatial_%d/*/*counter_collection.csv" % v)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "FETCH_SIZE": continue
        k = r["Kernel_Name"][:40]
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, (n, s) in acc.items():
        if "k_spmv<0, 1" in k or "k_spmv<0, 0" in k or "k_ssor_color<" in k:
            print("spatial", v, k, "launches", n, "fetch per launch %.1f MB (x2 KiB units)" % (2 * s * 1024 / n / 1e6))
PY
