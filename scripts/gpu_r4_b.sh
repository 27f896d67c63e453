#!/bin/bash
# Round 4, call B: whole GPU tier at the new defaults (centred fold, last layer on pooled rows, pipelined u8 staging) + bench A/B of the pruning.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r04_lnfold_centre_sweep.txt gpurun_out/r04_lnfold_centre_e2e.txt
echo "== GPU tests" ; timeout 1800 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -12 | tee gpurun_out/r04b_tests.log
echo "== bench (default)" ; timeout 900 python bench.py --json-out gpurun_out/r04b_bench.json 2>&1 | tail -1 | cut -c1-1600
echo "== bench, last layer on every row (A/B)" ; CLIP_AMD_PRUNE_LAST=0 timeout 600 python bench.py --no-matrix --no-cpu-baseline --no-host-api --json-out gpurun_out/r04b_bench_noprune.json 2>&1 | tail -1 | cut -c1-700
echo "== config cells" ; for c in cfg2_b32_q4_0_b32_img cfg3_l14_f16_b256_img; do timeout 600 python bench.py --config $c --no-cpu-baseline --no-host-api --json-out gpurun_out/r04b_$c.json 2>&1 | tail -1 | cut -c1-500; done
