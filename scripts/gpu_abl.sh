#!/bin/bash
# gemm8 ablation: dbg bit 0 no DMA in the loop, bit 1 no MFMA, bit 2 no fragment reads (variant library built with -DCLIPAMD_ABLATION)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-abl}
CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_abl.so timeout 600 python scripts/gemm_bench.py f16 160256 dbg0 dbg1 dbg2 dbg4 dbg3 dbg5 dbg6 dbg7 b32.down b32.up l14.down 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_gemm8_ablation.log
