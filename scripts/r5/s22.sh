#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05u
Q="--vision-only --no-matrix --no-cpu-baseline --no-host-api --no-roofline --no-rates --steps 200"
run() { echo -n "batch $1 $2: "; CLIP_AMD_TILE_OVERRIDE="$3" timeout 300 python bench.py --batch $1 $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run 112 "qkv 160, rest old     " "5600,2304,768,160128"
run 112 "all old (qkv 128)     " "5600,2304,768,128128"
run 176 "qkv 192, up old 160   " "8800,2304,768,192128;8800,3072,768,160128"
run 176 "all old (160, 160)    " "8800,2304,768,160128;8800,3072,768,160128"
run 224 "qkv 192               " "11200,2304,768,192128"
run 224 "qkv 160 (old)         " "11200,2304,768,160128"
run 256 "qkv 192, up 160 (old) " "12800,2304,768,192128;12800,3072,768,160128"
run 256 "qkv 192, up 192       " "12800,2304,768,192128;12800,3072,768,192128"
run 256 "all old               " "12800,2304,768,160128;12800,3072,768,160128"
run 448 "qkv 192, out old 160  " "22400,2304,768,192128;22400,768,768,160128"
run 448 "all old (160, 160)    " "22400,2304,768,160128;22400,768,768,160128"
done 2>&1 | tee gpurun_out/${T}_qkv_only_other_batches.txt
