#!/bin/bash
# BASELINE.json configs (single-GPU forms) through bench.py; one JSON line each under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-cfg}
run() { name=$1; shift; echo "== $name"; timeout 900 python bench.py "$@" --json-out gpurun_out/${TAG}_$name.json 2>&1 | tail -1 | cut -c1-900; }
run b32_q4_0_b32 --model b32 --ftype q4_0 --batch 32 --vision-only --steps 30 --no-cpu-baseline
run b32_q4_0_b1 --model b32 --ftype q4_0 --batch 1 --vision-only --steps 50 --no-cpu-baseline
run b32_q4_0_b256_vis --model b32 --ftype q4_0 --batch 256 --vision-only --steps 20 --no-cpu-baseline
run l14_f16_b256 --model l14 --ftype f16 --batch 256 --vision-only --steps 5 --warmup 2 --no-cpu-baseline
run l14_q5_1_b128 --model l14 --ftype q5_1 --batch 128 --vision-only --steps 5 --warmup 2 --no-cpu-baseline
run h14_q8_0_b64 --model h14 --ftype q8_0 --batch 64 --vision-only --steps 5 --warmup 2 --no-cpu-baseline
