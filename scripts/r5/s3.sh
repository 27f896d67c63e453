#!/bin/bash
# round 5, GPU session 3: where the time of the persistent 8-wave kernel goes (ablation build: p.debug switches parts of the kernel off)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05c
export CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_abl.so
{
echo "# gemm8p (tile 160257) ablation, q4_0 weight pre-dequantised (panel), fold form; dbg bits: 1 no LDS-DMA in the loop, 2 no MFMA, 4 no fragment reads, 8 no stores, 16 no epilogue arithmetic"
for it in 20 200; do
echo "## GEMM_ITERS=$it"
GEMM_ITERS=$it timeout 900 python scripts/gemm_bench.py q4_0 pre fold 160257 dbg0 dbg8 dbg16 dbg24 dbg1 dbg4 dbg5 dbg29 dbg2 dbg31 b32.qkv b32.up txt.up 2>&1 | grep -v amdgpu.ids
done
echo "# the same shapes without the fold (plain bias epilogue)"
GEMM_ITERS=200 timeout 900 python scripts/gemm_bench.py q4_0 pre 160257 dbg0 dbg8 dbg24 dbg29 b32.qkv b32.up 2>&1 | grep -v amdgpu.ids
echo "# gemm8 (160256; bits: 1 no DMA, 2 no MFMA, 4 no fragment reads) and the fused 4-wave kernel (160128; 1 no tile loads, 2 no MFMA, 4 no dequant-store, 8 no epilogue)"
GEMM_ITERS=200 timeout 900 python scripts/gemm_bench.py q4_0 pre fold 160256 dbg0 dbg1 dbg4 dbg5 dbg2 b32.qkv b32.up 2>&1 | grep -v amdgpu.ids
GEMM_ITERS=200 timeout 900 python scripts/gemm_bench.py q4_0 fold 160128 dbg0 dbg8 dbg1 dbg4 dbg2 b32.qkv b32.up 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/${T}_gemm8p_ablation.txt
