#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== tests" ; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "resident or config or large or model_shape" 2>&1 | tail -4
for m in 1 0 1 0; do
echo "== default config, CLIP_AMD_RESIDENT_PANELS=$m"
CLIP_AMD_RESIDENT_PANELS=$m timeout 600 python bench.py --no-matrix --no-cpu-baseline --no-host-api --no-roofline 2>&1 | tail -1 | cut -c1-140
done
for m in 1 0; do
echo "== cfg4 (ViT-L/14 q5_1, 128 images), CLIP_AMD_RESIDENT_PANELS=$m"
CLIP_AMD_RESIDENT_PANELS=$m timeout 900 python bench.py --config cfg4_l14_q5_1_b128_img --no-host-api 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('cpu_baseline') or {}).get('gpu_vs_cpu_1_minus_cos_max')); [print('   ', k, v) for k,v in list(d['kernels'].items())[:8]]"
done
