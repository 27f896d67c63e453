#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05t
Q="--vision-only --no-matrix --no-cpu-baseline --no-host-api --no-roofline --no-rates --steps 200"
run() { echo -n "batch $1 $2: "; CLIP_AMD_TILE_OVERRIDE="$3" timeout 300 python bench.py --batch $1 $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run 96  "new (up 192x128)      " ""
run 96  "old (up 128x128)      " "4800,3072,768,128128"
run 128 "new (qkv 160x128)     " ""
run 128 "old (qkv 128x128)     " "6400,2304,768,128128"
run 176 "new (qkv, up 192x128) " ""
run 176 "old (qkv, up 160x128) " "8800,2304,768,160128;8800,3072,768,160128"
run 400 "new (out 160x128)     " ""
run 400 "old (out 128x128)     " "20000,768,768,128128"
run 512 "new (out 192x128)     " ""
run 512 "old (out 160x128)     " "25600,768,768,160128"
done 2>&1 | tee gpurun_out/${T}_cost_model_other_batches.txt
