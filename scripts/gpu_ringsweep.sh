#!/bin/bash
# ring kernel vs the heuristic over mid-M shapes (text tower h=512 / 768, ViT-B/32, ViT-L/14): input for pick_tile
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-ringsweep}
T="0 65064 2065064 65128 2065128 3065128 65256"
for M in 130 257 514 640 1280 2560; do
  SH=""
  for nk in 1536x512:1 512x512:4 2048x512:3 512x2048:4 2304x768:1 768x768:4 3072x768:3 768x3072:4 3072x1024:1 1024x1024:4 4096x1024:3 1024x4096:4; do SH="$SH ${M}x${nk}"; done
  GEMM_ITERS=100 timeout 300 python scripts/gemm_bench.py q4_0 $T $SH 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/${TAG}_q4_0.log
for M in 257 514 1280; do
  SH=""
  for nk in 3072x1024:1 1024x1024:4 4096x1024:3 1024x4096:4 2304x768:1 768x768:4 3072x768:3 768x3072:4; do SH="$SH ${M}x${nk}"; done
  GEMM_ITERS=100 timeout 300 python scripts/gemm_bench.py f16 0 65064 2065064 65128 2065128 3065128 $SH 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/${TAG}_f16.log
