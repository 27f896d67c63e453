#!/bin/bash
# ViT-B/32 q4_0 batch-256 GEMM shapes, sustained (1000 launches, 12 rotating weight copies): fused-dequant 4-wave kernel vs the panel kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-b32sus}
SH="b32.qkv b32.up b32.out b32.down txt.qkv txt.up txt.out txt.down"
GEMM_ITERS=1000 GEMM_ROTATE=12 timeout 300 python scripts/gemm_bench.py q4_0 0 160128 192128 128128 $SH 2>&1 | grep -v amdgpu.ids | sed 's/^/fused /' | tee gpurun_out/${TAG}.log
GEMM_ITERS=1000 GEMM_ROTATE=12 timeout 300 python scripts/gemm_bench.py q4_0 pre 160128 160256 256256 256259 128256 $SH 2>&1 | grep -v amdgpu.ids | sed 's/^/panel /' | tee -a gpurun_out/${TAG}.log
