#!/bin/bash
# round 5, GPU session 2: the persistent 8-wave kernel (k_gemm8p.hip) — kernel tests, isolated timings against the fused 4-wave and the
# 8-wave kernels on a dequantised panel (fold form), same-box A/B of the default bench line; attention after the register rework
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05b
echo "== gemm8p + attention kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm8p or attention" 2>&1 | tail -8 | tee gpurun_out/${T}_tests.log
echo "== isolated timings (us), fold form, weight pre-dequantised for the panel kernels"
for ty in q4_0 f16; do
GEMM_ITERS=100 timeout 600 python scripts/gemm_bench.py $ty pre fold 0 160128 192128 160256 160257 b32.qkv b32.up txt.qkv txt.up l14.b32.qkv l14.b32.up b128.qkv b128.up 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/${T}_gemm8p_isolated.txt
Q="--no-matrix --no-cpu-baseline --no-host-api"
for rep in 1 2; do for on in 0 1; do
  echo "== bench CLIP_AMD_GEMM8P=$on (run $rep)"; CLIP_AMD_GEMM8P=$on timeout 300 python bench.py $Q --json-out gpurun_out/${T}_bench_8p${on}_$rep.json 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu']); k=d['kernels']
for n,v in list(k.items())[:12]: print('   %-70s %8.4f ms/step x%d  %s TF' % (n, v['ms_per_step'], v['launches_per_step'], v['tflops']))"
done; done 2>&1 | tee gpurun_out/${T}_bench_ab.txt
echo "== cfg3 (ViT-L/14 f16 batch 256)"; timeout 600 python bench.py --config cfg3_l14_f16_b256_img $Q --json-out gpurun_out/${T}_cfg3.json 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); k=d['kernels']
for n,v in list(k.items())[:10]: print('   %-70s %8.4f ms/step x%d  %s TF' % (n, v['ms_per_step'], v['launches_per_step'], v['tflops']))" | tee gpurun_out/${T}_cfg3.txt
