#!/bin/bash
# is the f16-panel advantage a short-burst effect?  same GEMM timed over 20 launches and over ~0.3 s, clocks sampled alongside
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-sus}
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/${TAG}_clocks.log 2>&1 &
SMI=$!
for it in 20 4000; do
  for mode in "" pre; do
    GEMM_ITERS=$it timeout 200 python scripts/gemm_bench.py q4_0 $mode 160128 b32.qkv b32.up b32.down txt.down 2>&1 | grep -v amdgpu.ids | sed "s/^/iters=$it $mode /" | tee -a gpurun_out/${TAG}_bench.log
  done
done
GEMM_ITERS=4000 timeout 200 python scripts/gemm_bench.py f16 160128 160256 256256 b32.up l14.up 2>&1 | grep -v amdgpu.ids | sed "s/^/iters=4000 /" | tee -a gpurun_out/${TAG}_bench.log
kill $SMI 2>/dev/null
cat gpurun_out/${TAG}_clocks.log | head -40
