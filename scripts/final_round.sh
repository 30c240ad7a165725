#!/bin/bash
# End-of-round validation on the GPU box (gpurun): full GPU test suite, smoke, the bench lines of configs 2, 3 and 5's solver,
# the nonlinear loop, kernel statistics.  Outputs under gpurun_out/final/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout -k 10 700 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -2 $OUT/pytest_gpu.log | tee -a $OUT/summary.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt; tail -1 $OUT/smoke.log | tee -a $OUT/summary.txt
timeout -k 10 300 python bench.py > $OUT/bench_10m_cg_ssor.json 2> $OUT/bench_10m_cg_ssor.err; echo "bench rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 300 python bench.py --elems 69 --precond 3 > $OUT/bench_1m_cg_diag.json 2> $OUT/bench_1m.err; echo "bench1m rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 300 python bench.py --method 2 --precond 10 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_10m_bicgstab_ilu0.json 2> $OUT/bench_ilu.err; echo "benchilu rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 300 python scripts/bench_nonlinear.py 149 1 1 3 2>/dev/null | tail -1 > $OUT/nonlinear_newton_10m.json; echo "nl rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 200 python scripts/bench_assembly.py 149 > $OUT/assembly.json 2>/dev/null
timeout -k 10 300 python scripts/bench_nn.py 1 150 > $OUT/nn_ndof1.json 2>/dev/null
timeout -k 10 300 python scripts/bench_nn.py 6 70 > $OUT/nn_ndof6.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline > $OUT/prof_run.json 2> $OUT/prof.err
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_cg_ssor_10m.csv
python3 - <<PY | tee -a $OUT/summary.txt
import json
for n in ("bench_10m_cg_ssor", "bench_1m_cg_diag", "bench_10m_bicgstab_ilu0"):
    try:
        d = json.loads(open("$OUT/%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"], 1), d["unit"][:20], "spmv frac", round(d["roofline"]["frac"], 3), "cpu", d.get("cpu_baseline", {}).get("value"))
    except Exception as e:
        print(n, "FAILED", e)
PY
