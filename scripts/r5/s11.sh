#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05k
Q="--no-matrix --no-cpu-baseline --no-host-api --no-roofline --no-rates --steps 300"
run() { echo -n "$1: "; CLIP_AMD_TILE_OVERRIDE="$2" timeout 300 python bench.py $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for rep in 1 2 3 4 5; do
run "heuristic          " ""
run "text qkv 192x128   " "10290,1536,192128"
run "both qkv 192x128   " "12800,2304,192128;10290,1536,192128"
run "text qkv 160x128   " "10290,1536,160128"
done 2>&1 | tee gpurun_out/${T}_tile_override_ab2.txt
