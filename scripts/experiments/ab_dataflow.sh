#!/bin/bash
# A/B of the triangular sweeps at 10.1M DOF: launch-per-level (FX_DATAFLOW=0) against the persistent dataflow launch.
# usage (GPU box): bash scripts/experiments/ab_dataflow.sh > gpurun_out/ab_dataflow.log 2>&1
set -o pipefail
cd "$(dirname "$0")/../.."
run() {  # label, env..., -- bench args
  local label=$1; shift
  echo "== $label"
  env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readline())
print("it/s %.1f  ms/it %.3f  precond_apply %.3f ms (%.0f GB/s)  spmv %.3f ms" % (d["value"], d["ms_per_step"], d["roofline"]["precond_apply"]["ms"], d["roofline"]["precond_apply"]["achieved_GBs"], d["roofline"]["ms_per_launch"]))'
}
ARGS="--method 2 --precond 10"
run "ILU levels"       FX_DATAFLOW=0 && \
run "ILU dataflow wps4 grid256" FX_DATAFLOW=1 && \
run "ILU dataflow wps8 grid256" FX_DATAFLOW=1 FX_DF_WPS=8 && \
run "ILU dataflow wps2 grid256" FX_DATAFLOW=1 FX_DF_WPS=2 && \
run "ILU dataflow wps4 grid512" FX_DATAFLOW=1 FX_DF_GRID=512 && \
run "ILU dataflow wps4 grid128" FX_DATAFLOW=1 FX_DF_GRID=128
ARGS="--method 1 --precond 1"
run "SSOR colours"     FX_DATAFLOW=0 && \
run "SSOR dataflow wps4 grid256" FX_DATAFLOW=2 && \
run "SSOR dataflow wps4 grid512" FX_DATAFLOW=2 FX_DF_GRID=512 && \
run "SSOR dataflow wps4 grid768" FX_DATAFLOW=2 FX_DF_GRID=768
