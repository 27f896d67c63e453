#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-p1}
echo "== p1 tests"; CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_p1.so timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "gemm8 or bitwise" 2>&1 | tail -3
echo "== base (2 phases, DMA first)"; timeout 300 python scripts/gemm_bench.py f16 160256 b32.qkv b32.out b32.up b32.down l14.up l14.down txt.qkv 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_base.log
echo "== p1"; CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_p1.so timeout 300 python scripts/gemm_bench.py f16 160256 b32.qkv b32.out b32.up b32.down l14.up l14.down txt.qkv 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_p1.log
echo "== p1 ablation"; CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_p1abl.so timeout 300 python scripts/gemm_bench.py f16 160256 dbg0 dbg1 dbg2 dbg4 dbg3 dbg5 dbg6 dbg7 b32.down b32.up 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_p1abl.log
