#!/bin/bash
# fistr1 itself on bench.py's workload (linear-elastic cube, CG + SSOR) at size: the unmodified program on every host core
# (oracle/_ref/fistr1_ref) against oracle/_ref/fistr1_hip.  usage: fistr1_linear.sh N [THREADS]   (N elements per edge)
R=$GRAFT_REPO_ROOT; N=${1:-69}; TH=${2:-16}; OUT=$R/gpurun_out/r4/fistr1_linear_$N; mkdir -p $OUT
D=/tmp/f1lin_$N; python3 $R/scripts/fistr1_cube_deck.py $D $N --linear | tee $OUT/deck.txt
cd $D
for bin in ${BINS:-fistr1_hip fistr1_hip_hostasm fistr1_ref}; do
  T0=$(date +%s.%N)
  if [ $bin = fistr1_hip_hostasm ]; then export HECMW_GPU_ASSEMBLY=0; exe=fistr1_hip; else unset HECMW_GPU_ASSEMBLY; exe=$bin; fi
  HECMW_GPU_REPORT=1 OMP_NUM_THREADS=$TH timeout -k 10 1000 $R/oracle/_ref/$exe > $OUT/stdout_$bin.txt 2>&1
  T1=$(date +%s.%N)
  echo "== $bin ($TH host threads) wall $(python3 -c "print('%.1f' % ($T1 - $T0))") s" | tee -a $OUT/summary.txt
  grep "BLOCK\|iterations\|set-up time\|solver time\|TOTAL TIME\|pre (sec)\|solve (sec)\|libfistr_hip" $OUT/stdout_$bin.txt | tee -a $OUT/summary.txt
  grep "U1\|U3" 0.log | tail -4 | tee -a $OUT/summary.txt
done
