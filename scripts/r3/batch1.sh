#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r3; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fistr1.py tests/test_gpu_fortran_shim.py tests/test_gpu_distributed.py -x -q > $OUT/t1.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/t1.log
FX_EISENSTAT=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/t1_eis.log 2>&1; echo "eis tests rc=$?"; tail -3 $OUT/t1_eis.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/b_std.json 2> $OUT/b_std.err && \
timeout -k 10 300 python bench.py --no-cpu-baseline --eisenstat > $OUT/b_eis.json 2> $OUT/b_eis.err && \
FX_EIS_FUSE=0 timeout -k 10 300 python bench.py --no-cpu-baseline --eisenstat > $OUT/b_eis_nofuse.json 2> $OUT/b_eis_nofuse.err
python3 - <<PY
import json
for f in ("b_std","b_eis","b_eis_nofuse"):
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); print(f, "%.1f it/s %.3f ms spmv %.3f prec %.3f ms"%(d["value"],d["ms_per_step"],d["roofline"]["ms_per_launch"],d["roofline"]["precond_apply"]["ms"]))
    except Exception as e: print(f,"FAILED",e)
PY
