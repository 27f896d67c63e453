#!/bin/bash
# round 5, GPU session 1: GPU test tier (new: configs 3/4/5 at benchmarked batch vs the oracle, split-graph vs pair call), default bench line
# with the matrix oracle deltas, batch split over 2 / 3 streams at batch 256 (CLIP_AMD_SPLIT), tile sweep of the eight BASELINE GEMM shapes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05a
echo "== GPU tests"; timeout 1500 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/${T}_tests.log
echo "== bench (default)"; timeout 900 python bench.py --json-out gpurun_out/${T}_bench.json 2>&1 | tail -1 | cut -c1-1500
Q="--no-matrix --no-cpu-baseline --no-roofline --no-host-api"
for sp in "0,0" "2,100000,2" "2,100000,3"; do
  echo "== bench CLIP_AMD_SPLIT=$sp"; CLIP_AMD_SPLIT=$sp timeout 300 python bench.py $Q 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu'])"
done 2>&1 | tee gpurun_out/${T}_split_b256.txt
echo "== tile sweep"; GEMM_ITERS=100 timeout 600 python scripts/gemm_bench.py q4_0 fold 0 128128 160128 192128 128064 64128 b32.qkv b32.out b32.up b32.down txt.qkv txt.out txt.up txt.down 2>&1 | tee gpurun_out/${T}_tile_sweep.txt
