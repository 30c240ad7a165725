#!/bin/bash
# hipGraph replay of the Krylov iteration (FX_GRAPH) on a launch-bound size: 32^3 nodes (98k DOF).
R=${GRAFT_REPO_ROOT:-.}
for g in 0 1; do
  for cfg in "1 3" "1 1" "2 1"; do
    set -- $cfg
    export FX_GRAPH=$g
    python3 $R/bench.py --elems 31 --method $1 --precond $2 --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys,os
j=json.loads(sys.stdin.readline())
print('graph', os.environ['FX_GRAPH'], 'method', $1, 'precond', $2, '%.0f it/s' % j['value'], '%.1f us/iter' % (1e3 * j['ms_per_step']), 'resid', j['resid_after_steps'])"
  done
done
