#!/bin/bash
# 4-wave 128x128-per-wave kernel (k_gemm4.hip): correctness, then A/B (burst and sustained)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-g4}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm8 or tiles_are_bitwise or whole_rounds" 2>&1 | tail -5 | tee gpurun_out/${TAG}_tests.log
timeout 400 python scripts/gemm_bench.py f16 160128 160256 256256 256259 256260 l14.up l14.down l14.qkv l14.out b32.qkv b32.up b32.out b32.down b1024.qkv b1024.up b1024.down txt.qkv txt.up 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_bench.log
GEMM_ITERS=1500 GEMM_ROTATE=4 timeout 400 python scripts/gemm_bench.py f16 160128 256258 256260 l14.up l14.down l14.qkv l14.out 2>&1 | grep -v amdgpu.ids | sed 's/^/sustained /' | tee -a gpurun_out/${TAG}_bench.log
