#!/bin/bash
# A/B of the wave-split sweep: ILU(0) (BiCGSTAB) and SSOR (CG) at 10.1M DOF.
# usage: ab_split.sh [split_max_slices:wps ...]
R=${GRAFT_REPO_ROOT:-.}
V=${@:-"0:4 2048:4 2048:2 8192:4"}
for v in $V; do
  export FX_SPLIT_MAX_SLICES=${v%%:*} FX_SPLIT_WPS=${v##*:}
  for cfg in "2 10 6" "1 1 40"; do
    set -- $cfg
    python3 $R/bench.py --method $1 --precond $2 --steps $3 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys,os
j=json.loads(sys.stdin.readline())
print('split', os.environ['FX_SPLIT_MAX_SLICES'], 'wps', os.environ['FX_SPLIT_WPS'], 'method', $1, 'precond', $2, 'it/s %.2f' % j['value'], 'apply ms %.3f' % j['roofline']['precond_apply']['ms'], 'spmv ms %.3f' % j['roofline']['ms_per_launch'])"
  done
done
