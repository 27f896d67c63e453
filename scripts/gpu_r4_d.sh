#!/bin/bash
# Round 4, call D: would resident fp16 panels for SOME weights (FFN-down: K = 3072) + the 8-wave kernel beat the fused-dequant 4-wave kernel in the layer?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for rot in 1 12; do
echo "== GEMM_ROTATE=$rot, fold epilogues, 200 iterations: fused 4-wave tiles vs 8-wave tiles on a pre-dequantised panel"
GEMM_ROTATE=$rot GEMM_ITERS=200 timeout 600 python scripts/gemm_bench.py b32.qkv b32.out b32.up b32.down txt.qkv txt.out txt.up txt.down fold 160128 192128 2>&1 | grep -v amdgpu.ids
GEMM_ROTATE=$rot GEMM_ITERS=200 timeout 600 python scripts/gemm_bench.py b32.qkv b32.out b32.up b32.down txt.qkv txt.out txt.up txt.down fold pre 96256 128256 160256 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r04d_panel_vs_fused.txt
