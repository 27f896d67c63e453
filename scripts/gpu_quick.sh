#!/bin/bash
# quick A/B: kernel tests + gemm microbench (+ optional full bench)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-q}; shift
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -3
echo "== gemm bench"; timeout 600 python scripts/gemm_bench.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_gemm_bench.log
