#!/bin/bash
# Round 2, call A: new parity tests + the 8-wave GEMM (tests, micro-benchmark vs the 4-wave tiles) + bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02a}
echo "== kernel tests" ; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -25 | tee gpurun_out/${TAG}_kernels.log
echo "== parity tests" ; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -25 | tee gpurun_out/${TAG}_parity.log
echo "== gemm bench (us / TF): f16 and q4_0, 4-wave tiles vs 8-wave tiles"
timeout 600 python scripts/gemm_bench.py f16 q4_0 160128 192128 160256 128256 96256 b32.qkv b32.out b32.up b32.down txt.qkv txt.out txt.up txt.down l14.up l14.down l14.qkv l14.out 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_gemm_bench.log
echo "== gemm bench, pre-dequantised panel (per-layer form)"
timeout 600 python scripts/gemm_bench.py pre q4_0 160256 128256 b32.qkv b32.out b32.up b32.down txt.qkv txt.up l14.up l14.down 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_gemm_bench_pre.log
echo "== bench" ; timeout 900 python bench.py --json-out gpurun_out/${TAG}_bench.json 2>&1 | tail -2 | cut -c1-6000 | tee gpurun_out/${TAG}_bench.log
