#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05f
{
echo "# 256 x 256 kernels on the text-tower shapes (weight pre-dequantised, plain epilogue)"
GEMM_ITERS=200 timeout 900 python scripts/gemm_bench.py q4_0 pre 0 160128 192128 256259 256260 256256 256258 txt.qkv txt.up txt.out txt.down 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/${T}_text_256.txt
