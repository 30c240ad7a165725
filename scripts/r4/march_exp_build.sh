#!/bin/bash
# Build the timing-experiment variants of the library used by scripts/r4/march_exp.sh (here, hipcc cross-compiles; the .so files travel with
# gpurun, they are git-ignored).  Each variant switches ONE thing of k_tri_march off (results are wrong on purpose, only the time is read):
#   NOPOLL never waits for a far entry | FARPLAIN far gathers as ordinary cached loads | NOFAR no far gathers | NOVALS one value word instead
#   of nine | NOVALSNOFAR both
cd "$(dirname "$0")/../../frontistr_amd/csrc" || exit 1
mkdir -p ../../scripts/r4/libs
for v in "NOPOLL:-DFX_MARCH_EXP_NOPOLL" "FARPLAIN:-DFX_MARCH_EXP_NOPOLL -DFX_MARCH_EXP_FARPLAIN" "NOFAR:-DFX_MARCH_EXP_NOPOLL -DFX_MARCH_EXP_NOFAR" \
         "NOVALS:-DFX_MARCH_EXP_NOPOLL -DFX_MARCH_EXP_NOVALS" "NOVALSNOFAR:-DFX_MARCH_EXP_NOPOLL -DFX_MARCH_EXP_NOVALS -DFX_MARCH_EXP_NOFAR"; do
  n=${v%%:*}; f=${v#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result $f -I/opt/rocm/include -c fistr_hip.hip -o /tmp/fx_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../scripts/r4/libs/libfx_$n.so /tmp/fx_$n.o fx_order.o -L/opt/rocm/lib -ldl -lpthread -Wl,-rpath,/opt/rocm/lib && echo built $n ) &
done
wait
