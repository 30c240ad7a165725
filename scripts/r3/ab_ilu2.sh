#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3; cd $R
B="FX_DF_SOA=1,FX_DF_GRID=0,FX_DF_WPS=8,FX_DF_SLEEP=2,FX_DF_POLL=1"
AB_BASE=$B python scripts/r3/ab_opts.py --method 2 --precond 10 "FX_DF_GRID=256" "FX_DF_GRID=256,FX_DF_SLEEP=0" "FX_DF_GRID=256,FX_DF_POLL=0" "FX_DF_GRID=384" "FX_DF_GRID=512" 2>&1 | grep "^rep 2" | sed 's/^/cached   /'
FX_DF_UNCACHED=1 AB_BASE=$B python scripts/r3/ab_opts.py --method 2 --precond 10 "FX_DF_GRID=256" "FX_DF_GRID=256,FX_DF_SLEEP=0" 2>&1 | grep "^rep 2" | sed 's/^/uncached /'
