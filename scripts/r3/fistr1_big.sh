#!/bin/bash
# fistr1 itself at size: the synthetic NLSTATIC cube (scripts/fistr1_cube_deck.py) through oracle/_ref/fistr1_hip with the
# element loops on the device (default) and on the host (HECMW_GPU_ASSEMBLY=0).  usage: fistr1_big.sh N [SUBSTEPS] [STRAIN]
R=$GRAFT_REPO_ROOT; N=${1:-60}; SUB=${2:-2}; STRAIN=${3:-0.004}; OUT=$R/gpurun_out/r3/fistr1_big_$N; mkdir -p $OUT
for mode in device host; do
  D=/tmp/f1big_$mode; rm -rf $D; python3 $R/scripts/fistr1_cube_deck.py $D $N $SUB CG 1 $STRAIN > /dev/null
  cd $D
  if [ $mode = host ]; then export HECMW_GPU_ASSEMBLY=0; else unset HECMW_GPU_ASSEMBLY; fi
  T0=$(date +%s.%N)
  timeout -k 10 900 $R/oracle/_ref/fistr1_hip > $OUT/stdout_$mode.txt 2>&1
  T1=$(date +%s.%N)
  cp FSTR.sta $OUT/FSTR_$mode.sta 2>/dev/null
  echo "$mode wall $(python3 -c "print('%.1f' % ($T1 - $T0))") s" | tee $OUT/time_$mode.txt
  grep -c "3x3 BLOCK" $OUT/stdout_$mode.txt
  grep "set-up time" $OUT/stdout_$mode.txt | head -4
  grep "TOTAL TIME\|solve (sec)" $OUT/stdout_$mode.txt
done
cd $R
python3 - <<PY
import re
for mode in ("device", "host"):
    t = open("$OUT/stdout_%s.txt" % mode).read()
    its = re.findall(r"iter:\s+(\d+), residual: (\S+), disp.corr.: (\S+)", t)
    print(mode, "Newton lines", len(its), its[:3], "...", its[-2:])
PY
