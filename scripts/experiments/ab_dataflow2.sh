#!/bin/bash
# grid / waves-per-slice sweep of the ILU(0) dataflow sweep at 10.1M DOF
set -o pipefail
cd "$(dirname "$0")/../.."
run() {
  local label=$1; shift
  echo "== $label"
  env "$@" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --method 2 --precond 10 2>/dev/null | python3 -c '
import json,sys
d=json.loads(sys.stdin.readline())
print("it/s %.1f  ms/it %.3f  precond_apply %.3f ms (%.0f GB/s)  spmv %.3f ms" % (d["value"], d["ms_per_step"], d["roofline"]["precond_apply"]["ms"], d["roofline"]["precond_apply"]["achieved_GBs"], d["roofline"]["ms_per_launch"]))'
}
for g in 96 128 192; do
  for w in 4 8; do
    run "ILU dataflow wps$w grid$g" FX_DATAFLOW=1 FX_DF_WPS=$w FX_DF_GRID=$g || exit 1
  done
done
