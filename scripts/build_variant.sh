#!/bin/bash
# Kernel A/B tooling: build clip_cpp_amd/variants/libclip_<NAME>.so with extra -D defines applied to the GEMM translation
# units (all other objects are reused from clip_cpp_amd/build).  Load it with CLIP_AMD_LIB=<path> (clip_cpp_amd.lib()).
# usage: scripts/build_variant.sh NAME -DCLIPAMD_SCHED=1 ...
set -eu
cd "$(dirname "$0")/.."
NAME=$1; shift
python -m clip_cpp_amd.build > /dev/null
B=clip_cpp_amd/build; V=clip_cpp_amd/variants/$NAME; mkdir -p $V
for wt in 0 1 2 3 4 5; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude "$@" -DCLIPAMD_GEMM_WT=$wt -c clip_cpp_amd/csrc/k_gemm.hip -o $V/k_gemm_wt$wt.o &
done
for wt in 0 1 2 3 4 5; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -Iinclude "$@" -DCLIPAMD_RING_WT=$wt -c clip_cpp_amd/csrc/k_gemm_ring.hip -o $V/k_gemm_ring_wt$wt.o &
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -Iinclude "$@" -c clip_cpp_amd/csrc/k_gemm_ring.hip -o $V/k_gemm_ring.hip.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude "$@" -c clip_cpp_amd/csrc/k_gemm.hip -o $V/k_gemm.hip.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude "$@" -c clip_cpp_amd/csrc/k_gemm8.hip -o $V/k_gemm8.hip.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude "$@" -c clip_cpp_amd/csrc/k_gemm4.hip -o $V/k_gemm4.hip.o &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -Iinclude "$@" -c clip_cpp_amd/csrc/k_gemm32.hip -o $V/k_gemm32.hip.o &
wait
OBJS=$(ls $B/*.o | grep -v 'k_gemm_wt\|k_gemm\.hip\|k_gemm_ring\|k_gemm8\|k_gemm4\|k_gemm32')
hipcc --offload-arch=gfx950 -shared -fPIC -o clip_cpp_amd/variants/libclip_$NAME.so $OBJS $V/*.o -lz -lpthread -ldl
echo built clip_cpp_amd/variants/libclip_$NAME.so
