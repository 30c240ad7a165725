#!/bin/bash
# Bisect a crash of fistr1_hip on the linear cube deck: variants of the environment, exit code + last lines of each.
R=$GRAFT_REPO_ROOT; N=${1:-99}; D=/tmp/f1b_$N
python3 $R/scripts/fistr1_cube_deck.py $D $N --linear > /dev/null
sed -i 's/ITERLOG=NO/ITERLOG=YES/' $D/cube.cnt
cd $D
ulimit -s
for v in "A:" "B:FX_ARENA_TRIES=1" "C:HECMW_GPU_UPDATE=0" "D:FX_ARENA_GB=0" "E:HECMW_GPU_ASSEMBLY=0"; do
  tag=${v%%:*}; kv=${v#*:}
  env HECMW_GPU_REPORT=1 OMP_NUM_THREADS=16 $kv timeout -k 10 600 $R/oracle/_ref/fistr1_hip > $R/gpurun_out/r4/f1b_$tag.txt 2>&1
  echo "== variant $tag ($kv) exit $?"
  grep -c "^ *[0-9]* *[0-9]\.[0-9]*E" $R/gpurun_out/r4/f1b_$tag.txt
  tail -4 $R/gpurun_out/r4/f1b_$tag.txt
done
