#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in 0 1 5 0 5; do
echo "== CLIP_AMD_FF2_PANEL=$m"
CLIP_AMD_FF2_PANEL=$m timeout 600 python bench.py --no-matrix --no-cpu-baseline --no-host-api 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'img/s', d['images_per_s_per_gpu'], 'txt/s', d['texts_per_s_per_gpu']); [print('   ', k, v) for k,v in list(d['kernels'].items())[:8]]"
done
