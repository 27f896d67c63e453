#!/bin/bash
# Round 4, call C: the batch split over two streams (tests + sweep), the tests touched since call B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests" ; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pooled or split or fold" 2>&1 | tail -12 | tee gpurun_out/r04c_tests.log
echo "== split sweep" ; timeout 900 python scripts/split_bench.py b32:q4_0 l14:f16 2>&1 | grep -v "^$" | tee gpurun_out/r04c_split_sweep.txt
echo "== cfg2 + b32 matrix cell with the split at its default"
for c in cfg2_b32_q4_0_b32_img b32_q4_0_b32; do timeout 600 python bench.py --config $c --no-cpu-baseline --no-host-api --no-roofline 2>&1 | tail -1 | cut -c1-420; done
