#!/bin/bash
# Round 3: tile sweep for 4500-8000 rows (batches of ~100-200 texts, 96-160 ViT-B/32 images): where pick_tile's cost model is off
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for M in 4500 6000 8000; do
  GEMM_ITERS=100 timeout 300 python scripts/gemm_bench.py q4_0 0 64128 128128 160128 192128 2192128 65128 ${M}x512x512:4 ${M}x512x2048:4 ${M}x1536x512:1 ${M}x2048x512:3 ${M}x768x768:4 ${M}x768x3072:4 ${M}x2304x768:1 ${M}x3072x768:3 2>&1 | grep -v amdgpu.ids | cut -c1-420
done | tee gpurun_out/r03_midlarge_sweep.txt
