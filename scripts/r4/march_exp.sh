#!/bin/bash
# timing experiments on the plane march at 10.1 M DOF: which part of a round costs what (libraries built by scripts/r4/march_exp_build.sh with -DFX_MARCH_EXP_*; results of those are wrong on purpose)
mkdir -p gpurun_out/r4
for v in "" NOPOLL FARPLAIN NOFAR NOVALS NOVALSNOFAR; do
  echo "##### variant ${v:-product}"
  if [ -n "$v" ]; then export FX_LIBPATH=$PWD/scripts/r4/libs/libfx_$v.so; else unset FX_LIBPATH; fi
  timeout -k 10 200 python3 scripts/r4/march_bench.py "$@" 2>&1 | grep -v "^==" || exit 1
done
