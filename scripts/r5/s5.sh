#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05e
{
echo "# whole-rounds split of the 256 x 256 kernels (tile codes 256258 / 256260) on the ViT-B/32 batch-256 shapes, weight pre-dequantised"
GEMM_ITERS=200 timeout 900 python scripts/gemm_bench.py q4_0 pre 0 256259 256260 256256 256258 b32.qkv b32.up b32.out b32.down 2>&1 | grep -v amdgpu.ids
GEMM_ITERS=200 timeout 900 python scripts/gemm_bench.py q4_0 pre fold 0 256260 256258 b32.qkv b32.up b32.out b32.down 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/${T}_whole_rounds_b32.txt
