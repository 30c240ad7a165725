#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/full_gpu.log 2>&1; echo "rc=$?"; tail -15 $OUT/full_gpu.log
