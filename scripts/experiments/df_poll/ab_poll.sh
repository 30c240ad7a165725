#!/bin/bash
# dataflow sweeps, BiCGSTAB + ILU(0) at 10.1 M DOF: LDS-only barrier (new build) against __syncthreads (lib_base.so), and the
# staggered re-polls (FX_DF_POLL=2) on top of either
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3; cd $R
AB_BASE="FX_DF_POLL=1,FX_DF_SLEEP=0" bash scripts/r3/ab_builds.sh ab_poll scripts/r3/ab_opts.py --method 2 --precond 10 "FX_DF_POLL=2,FX_DF_SLEEP=3" "FX_DF_POLL=2,FX_DF_SLEEP=6"
