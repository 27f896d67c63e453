#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05n
echo "== fold kernel tests"; timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -k "lnfold or gemm4 or large_m or gemm8" 2>&1 | grep -E " passed| failed|Error|assert" | tail -5 | tee gpurun_out/${T}_tests.log
Q="--no-matrix --no-cpu-baseline --no-host-api --no-roofline --no-rates"
for rep in 1 2; do for f in 1 2; do
  echo -n "cfg3 CLIP_AMD_LNFOLD=$f: "; CLIP_AMD_LNFOLD=$f timeout 600 python bench.py --config cfg3_l14_f16_b256_img $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/${T}_cfg3_fold_ab.txt
for f in 1 2; do
  echo -n "cfg4 CLIP_AMD_LNFOLD=$f: "; CLIP_AMD_LNFOLD=$f timeout 600 python bench.py --config cfg4_l14_q5_1_b128_img $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  echo -n "cfg5 CLIP_AMD_LNFOLD=$f: "; CLIP_AMD_LNFOLD=$f timeout 600 python bench.py --config cfg5_h14_q8_0_b64_img $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done 2>&1 | tee -a gpurun_out/${T}_cfg3_fold_ab.txt
