#!/bin/bash
# A/B of two builds on one box: the baseline library (scripts/r3/lib_base.so) and the current one, alternating processes.
# usage: ab_builds.sh <tag> <python script + args...>
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3; cd $R; TAG=$1; shift
for rep in 1 2; do
  FX_LIBPATH=$R/scripts/r3/lib_base.so python "$@" 2>&1 | grep "^rep 2" | sed "s/^/base $rep: /" | tee -a gpurun_out/r3/$TAG.log
  python "$@" 2>&1 | grep "^rep 2" | sed "s/^/new  $rep: /" | tee -a gpurun_out/r3/$TAG.log
done
