#!/bin/bash
# Round-2 measurement batch A (GPU box, via gpurun): bench lines of configs 2 / 3 / 5's solver, the measured full-size CPU
# baseline, the 2- and 4-rank rehearsal of the self-launching bench, the nonlinear loop.  Outputs under gpurun_out/r02/.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd $R
FX_TIMING=1 timeout -k 10 300 python bench.py > $OUT/bench_10m_cg_ssor.json 2> $OUT/bench_10m_cg_ssor.err && echo "bench ok" && \
timeout -k 10 300 python bench.py --eisenstat --no-cpu-baseline > $OUT/bench_10m_cg_ssor_eisenstat.json 2> $OUT/bench_eis.err && echo "bench eisenstat ok" && \
timeout -k 10 300 python bench.py --elems 69 --precond 3 > $OUT/bench_1m_cg_diag.json 2> $OUT/bench_1m.err && echo "bench1m ok" && \
timeout -k 10 300 python bench.py --method 2 --precond 10 --steps 40 --warmup 5 --no-cpu-baseline > $OUT/bench_10m_bicgstab_ilu0.json 2> $OUT/bench_ilu.err && echo "benchilu ok" && \
FX_BENCH_TRANSPORT=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 30 --warmup 5 > $OUT/bench_2rank_rehearsal.json 2> $OUT/bench_2rank.err && echo "2rank ok" && \
FX_BENCH_TRANSPORT=gloo FX_OVERLAP=0 timeout -k 10 400 python bench.py --gpus 2 --steps 30 --warmup 5 > $OUT/bench_2rank_rehearsal_nooverlap.json 2> $OUT/bench_2rank_no.err && echo "2rank nooverlap ok" && \
FX_BENCH_TRANSPORT=gloo timeout -k 10 500 python bench.py --gpus 4 --steps 30 --warmup 5 > $OUT/bench_4rank_rehearsal.json 2> $OUT/bench_4rank.err && echo "4rank ok" && \
timeout -k 10 300 python scripts/bench_nonlinear.py 149 1 1 3 2>/dev/null | tail -1 > $OUT/nonlinear_newton_10m.json && echo "nl ok" && \
timeout -k 10 300 python scripts/bench_nonlinear.py 149 2 10 3 2>/dev/null | tail -1 > $OUT/nonlinear_newton_10m_bicgstab_ilu0.json && echo "nl ilu ok" && \
timeout -k 10 200 python scripts/bench_assembly.py 149 > $OUT/assembly_10m.json 2>/dev/null && echo "asm ok" && \
timeout -k 10 300 python scripts/bench_nn.py 1 150 2>/dev/null | tail -1 > $OUT/nn_ndof1_3p4m.json && \
timeout -k 10 300 python scripts/bench_nn.py 6 70 2>/dev/null | tail -1 > $OUT/nn_ndof6_2m.json && \
timeout -k 10 300 python scripts/bench_nn.py 2 120 2>/dev/null | tail -1 > $OUT/nn_ndof2_3p5m.json && \
timeout -k 10 300 python scripts/bench_nn.py 5 70 2>/dev/null | tail -1 > $OUT/nn_ndof5_1p7m.json && echo "nn ok" && \
timeout -k 10 1100 python bench.py --steps 20 --warmup 5 --cpu-full > $OUT/bench_10m_cg_ssor_cpu_full.json 2> $OUT/bench_cpu_full.err && echo "cpu-full ok"
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), "value %.1f" % d["value"], "n_gpus", d["n_gpus"], "spmv frac %.3f" % d["roofline"]["frac"], "precond ms %.3f" % d["roofline"]["precond_apply"]["ms"], "cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("extrapolated"))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
