#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05m
Q="--config cfg2_b32_q4_0_b32_img --no-matrix --no-cpu-baseline --no-host-api --no-roofline --no-rates --steps 1000"
run() { echo -n "$1: "; CLIP_AMD_TILE_OVERRIDE="$2" CLIP_AMD_SPLIT="${3:-}" timeout 300 python bench.py $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_median'])"; }
for rep in 1 2; do
run "heuristic                  " ""
run "split 2 ways               " "" "2,64,2"
run "qkv 128x128                " "1600,2304,768,128128"
run "qkv ring 64x64             " "1600,2304,768,65064"
run "up ring 64x128             " "1600,3072,768,65128"
run "up 128x128                 " "1600,3072,768,128128"
run "up 192x128                 " "1600,3072,768,192128"
run "out ring 64x128            " "1600,768,768,65128"
run "out 64x64 split-K 2        " "1600,768,768,2064064"
run "down ring 64x64            " "1600,768,3072,65064"
run "down ring 64x128 no split  " "1600,768,3072,1065128"
run "down 64x128 split-K 4      " "1600,768,3072,4064128"
run "down 64x64 split-K 4       " "1600,768,3072,4064064"
done 2>&1 | tee gpurun_out/${T}_cfg2_tile_override.txt
