#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== attention tests" ; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -3
echo "== cfg3 (odd attention block rotated over the waves)"
timeout 600 python bench.py --config cfg3_l14_f16_b256_img --no-cpu-baseline --no-host-api --json-out gpurun_out/r04f_cfg3.json 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print('   ', k, v) for k,v in list(d['kernels'].items())[:7]]"
for sp in "0,0" ""; do
echo "== b32_q4_0_b32 two-tower cell, CLIP_AMD_SPLIT='$sp'"
if [ -n "$sp" ]; then export CLIP_AMD_SPLIT=$sp; else unset CLIP_AMD_SPLIT; fi
timeout 600 python bench.py --config b32_q4_0_b32 --no-cpu-baseline --no-host-api --no-roofline 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py --config l14_f16_b32 --no-cpu-baseline --no-host-api --no-roofline 2>&1 | tail -1 | cut -c1-200
done
