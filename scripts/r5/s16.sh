#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
Q="--no-matrix --no-cpu-baseline --no-host-api --no-rates"
for f in 1 2; do
  echo "== cfg3 CLIP_AMD_LNFOLD=$f"; CLIP_AMD_LNFOLD=$f timeout 600 python bench.py --config cfg3_l14_f16_b256_img $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); k=d['kernels']
for n,v in list(k.items())[:9]: print('   %-70s %8.4f ms/step x%d  %s TF' % (n, v['ms_per_step'], v['launches_per_step'], v['tflops']))"
done 2>&1 | tee gpurun_out/r05o_cfg3_fold_kernels.txt
