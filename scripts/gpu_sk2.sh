#!/bin/bash
# stream-K 256x256 kernel: correctness tests, then A/B against the fixed-tile forms
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-sk2}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "stream_k or gemm8" 2>&1 | tail -5 | tee gpurun_out/${TAG}_tests.log
timeout 400 python scripts/gemm_bench.py f16 160128 160256 256256 256257 b32.qkv b32.up b32.out b32.down l14.up l14.down l14.qkv l14.out txt.qkv txt.up txt.out txt.down b128.up b64.up 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_bench.log
timeout 300 python scripts/gemm_bench.py q4_0 pre 160128 160256 256257 b32.qkv b32.up b32.down l14.up 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_bench.log
