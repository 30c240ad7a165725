#!/bin/bash
# Fresh-process check of the SpMV's speed class: N processes of `bench.py --no-cpu-baseline` with the value arena
# (the default) and N with FX_ARENA_GB=0 (every value array its own power-of-two hipMalloc: rounds 2-3), alternating, on one box.
# One line per process: roofline.frac of the SpMV, its ms per launch, the headline and the other recurrence, where the arrays live.
R=$GRAFT_REPO_ROOT; N=${1:-5}; OUT=$R/gpurun_out/r4/ab_placement.txt
mkdir -p $R/gpurun_out/r4; cd $R
echo "# $(date -u +%FT%TZ)  bench.py --no-cpu-baseline --steps 40 --warmup 5, fresh processes alternating arena (default) / FX_ARENA_GB=0" >> $OUT
for i in $(seq 1 $N); do
  for mode in arena noarena; do
    if [ $mode = noarena ]; then export FX_ARENA_GB=0; else unset FX_ARENA_GB; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2> $R/gpurun_out/r4/ab_placement.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; p = r['placement']; v = d['variants']
o = list(v.values())[0] if v else {}
print('process %s  %-8s SpMV frac %.3f  %.4f ms (stream ceiling %.0f GB/s)  apply %.3f ms  headline %.1f it/s (%s)  other form %.1f it/s  arena %.0f GiB, SpMV values in it: %s; verification timed %d arena(s), kept %.4f ms'
      % ('$i', '$mode', r['frac'], r['ms_per_launch'], r['measured_stream_ceiling'], r['precond_apply']['ms'], d['value'], d['config']['recurrence'].split()[0],
         o.get('it_per_s', 0.0), p['arena_bytes'] / 2.0**30, p['spmv_values_in_arena'], p['arenas_timed'], p['kept_ms']))" >> $OUT || exit 1
  done
done
unset FX_ARENA_GB
cat $OUT
