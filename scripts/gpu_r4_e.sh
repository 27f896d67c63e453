#!/bin/bash
# Round 4, call E: LayerNorm fold inside the four-wave 256 x 256 kernel (ViT-L/14 f16 batch 256): tests + A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== lnfold kernel tests" ; timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "lnfold or large_m" 2>&1 | tail -4
for f in 1 2; do
echo "== cfg3 (ViT-L/14 f16, 256 images) CLIP_AMD_LNFOLD=$f"
CLIP_AMD_LNFOLD=$f timeout 600 python bench.py --config cfg3_l14_f16_b256_img --no-cpu-baseline --no-host-api --json-out gpurun_out/r04e_cfg3_fold$f.json 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); [print('   ', k, v) for k,v in list(d['kernels'].items())[:9]]"
done
echo "== gemm_bench fold vs plain, f16, L/14 shapes, tile 256259"
GEMM_ITERS=50 timeout 600 python scripts/gemm_bench.py f16 l14.qkv l14.out l14.up l14.down 256259 2>&1 | grep -v amdgpu.ids
GEMM_ITERS=50 timeout 600 python scripts/gemm_bench.py f16 l14.qkv l14.out l14.up l14.down 256259 fold 2>&1 | grep -v amdgpu.ids
