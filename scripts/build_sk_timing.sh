#!/bin/bash
# Timing build of the small-M path: clip_cpp_amd/variants/libclip_sktiming.so = the product objects with k_skinny.hip and forward.cpp
# recompiled with -DCLIPAMD_SK_TIMING (phase stamps of the first / last workgroup of every skinny launch; scripts/sk_stamps.py reads them).
set -eu
cd "$(dirname "$0")/.."
python -m clip_cpp_amd.build > /dev/null
B=clip_cpp_amd/build; V=clip_cpp_amd/variants/sktiming; mkdir -p $V
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -Iinclude -DCLIPAMD_SK_TIMING $*"
for wt in 0 1 2 3 4 5; do hipcc $F -DCLIPAMD_SKINNY_WT=$wt -c clip_cpp_amd/csrc/k_skinny.hip -o $V/k_skinny_wt$wt.o & done
hipcc $F -c clip_cpp_amd/csrc/k_skinny.hip -o $V/k_skinny.hip.o &
hipcc -x hip $F -c clip_cpp_amd/csrc/forward.cpp -o $V/forward.cpp.o &
wait
OBJS=$(ls $B/*.o | grep -v 'k_skinny\|forward.cpp')
hipcc --offload-arch=gfx950 -shared -fPIC -o clip_cpp_amd/variants/libclip_sktiming.so $OBJS $V/*.o -lz -lpthread -ldl
echo built clip_cpp_amd/variants/libclip_sktiming.so
