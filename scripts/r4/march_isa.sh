#!/bin/bash
# device ISA of k_tri_march<NW> (default 8): waits, barriers, register count -- the checks of fx_march.h's header
NW=${1:-8}
cd /root/repo/frontistr_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I/opt/rocm/include -S --cuda-device-only fistr_hip.hip -o /tmp/fistr_dev.s 2>&1 | grep -v "hip-link"
S=$(grep -n "^_Z11k_tri_marchILi${NW}EEv9MarchArgsPKi:" /tmp/fistr_dev.s | cut -d: -f1)
E=$(awk -v s=$S 'NR>s && /^\.Lfunc_end/ {print NR; exit}' /tmp/fistr_dev.s)
awk -v s=$S -v e=$E 'NR>=s && NR<=e' /tmp/fistr_dev.s > /tmp/m${NW}.s
grep -n "s_waitcnt vmcnt\|s_barrier\|Loop Header: Depth=2" /tmp/m${NW}.s
grep -n "num_vgpr\|private_seg_size" /tmp/fistr_dev.s | grep "marchILi${NW}E"
