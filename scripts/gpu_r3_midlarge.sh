#!/bin/bash
# Round 3: tile sweeps around pick_tile's regime boundaries.  usage: gpu_r3_midlarge.sh "<M list>" "<tile list>" OUT
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
MS=${1:-"4500 6000 8000"}; TILES=${2:-"0 64128 128128 160128 192128 2192128 65128"}; OUT=${3:-r03_midlarge_sweep}
for M in $MS; do
  GEMM_ITERS=100 timeout 300 python scripts/gemm_bench.py q4_0 $TILES ${M}x512x512:4 ${M}x512x2048:4 ${M}x1536x512:1 ${M}x2048x512:3 ${M}x768x768:4 ${M}x768x3072:4 ${M}x2304x768:1 ${M}x3072x768:3 2>&1 | grep -v amdgpu.ids | cut -c1-420
done | tee gpurun_out/$OUT.txt
