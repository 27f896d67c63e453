#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05d
{
echo "# vendor yardstick on this box (hipBLASLt through torch, plain f16 x f16 -> f16, no epilogue)"
timeout 600 python scripts/gemm_bench.py blas b32.qkv b32.up b32.out b32.down txt.qkv txt.up 2>&1 | grep -v amdgpu.ids
echo "# every large-tile kernel on the fp16-output shapes (weight pre-dequantised), plain bias epilogue and fold form"
GEMM_ITERS=200 timeout 900 python scripts/gemm_bench.py q4_0 pre 0 160128 160256 160257 256256 256259 b32.qkv b32.up 2>&1 | grep -v amdgpu.ids
GEMM_ITERS=200 timeout 900 python scripts/gemm_bench.py q4_0 pre fold 0 160128 160256 160257 256256 b32.qkv b32.up 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/${T}_large_tiles.txt
