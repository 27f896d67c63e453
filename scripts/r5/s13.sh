#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05l
Q="--no-matrix --no-cpu-baseline --no-host-api --no-roofline --no-rates --steps 300"
run() { echo -n "$1: "; CLIP_AMD_TILE_OVERRIDE="$2" timeout 300 python bench.py $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for rep in 1 2 3; do
run "heuristic                 " ""
run "vision out 192x128        " "12800,768,768,192128"
run "vision out 128x128        " "12800,768,768,128128"
run "vision down fused 160x128 " "12800,768,3072,160128"
run "vision down fused 192x128 " "12800,768,3072,192128"
run "vision up 160x128         " "12800,3072,768,160128"
run "text out 160x128          " "10290,512,512,160128"
run "text out 128x128          " "10290,512,512,128128"
run "text down 160x128         " "10290,512,2048,160128"
run "text down 128x128         " "10290,512,2048,128128"
run "text up 160x128           " "10290,2048,512,160128"
run "text up 128x128           " "10290,2048,512,128128"
done 2>&1 | tee gpurun_out/${T}_tile_override_ab3.txt
