#!/bin/bash
# level-scheduled sweeps: rows ordered by availability + the finishing wave taking the blocks that wait for the previous level
# (new build; FX_DF_LASTFIRST=0 = the sorted layout with the old wave mapping) against the previous build (scripts/r3/lib_base.so)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3; cd $R
for rep in 1 2; do AB_BASE="FX_DF_LASTFIRST=1" python scripts/r3/ab_opts.py --method 2 --precond 10 "FX_DF_LASTFIRST=0" 2>&1 | grep "^rep 2" | sed "s/^/new  $rep: /"; FX_LIBPATH=$R/scripts/r3/lib_base.so python scripts/r3/ab_opts.py --method 2 --precond 10 2>&1 | grep "^rep 2" | sed "s/^/base $rep: /"; done
