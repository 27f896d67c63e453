#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05j
Q="--no-matrix --no-cpu-baseline --no-host-api --no-roofline"
run() { echo "== $1"; CLIP_AMD_TILE_OVERRIDE="$2" timeout 300 python bench.py $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu'])"; }
for rep in 1 2; do
run "heuristic" ""
run "vision qkv 192x128" "12800,2304,192128"
run "vision qkv+up 192x128" "12800,2304,192128;12800,3072,192128"
run "text qkv 192x128" "10290,1536,192128"
run "vision out 128x128" "12800,768,128128"
run "vision qkv 192 + text qkv 192" "12800,2304,192128;10290,1536,192128"
done 2>&1 | tee gpurun_out/${T}_tile_override_ab.txt
