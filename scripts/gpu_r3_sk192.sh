#!/bin/bash
# Round 3: split-K on the 192 x 128 tile (text-tower long-K GEMMs) — tests + micro-benchmark, plain and fold form
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "split_k or lnfold" 2>&1 | grep -E "passed|failed" | tail -2
for m in "" fold; do echo "== gemm_bench [$m]"; GEMM_ITERS=200 timeout 300 python scripts/gemm_bench.py q4_0 $m 0 192128 2192128 3192128 4192128 txt.down txt.out txt.up txt.qkv 2>&1 | grep -v amdgpu.ids | cut -c1-400; done
echo "== f16"; GEMM_ITERS=200 timeout 300 python scripts/gemm_bench.py f16 0 192128 2192128 3192128 txt.down txt.out 2>&1 | grep -v amdgpu.ids | cut -c1-400
echo "== H/14 text + L/14 text shapes (N = 1024 / 768, K = 4096 / 3072)"; GEMM_ITERS=200 timeout 300 python scripts/gemm_bench.py q4_0 0 192128 2192128 3192128 10290x1024x4096:4 10290x768x3072:4 5000x512x2048:4 2>&1 | grep -v amdgpu.ids | cut -c1-400
