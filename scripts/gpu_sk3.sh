#!/bin/bash
# whole-rounds split of the 256x256 tile + resident fp16 panels: correctness, then A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-sk3}
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -5 | tee gpurun_out/${TAG}_tests.log
timeout 400 python scripts/gemm_bench.py f16 160128 160256 256256 256258 l14.up l14.down l14.qkv l14.out b1024.qkv b1024.out b1024.up b1024.down 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_bench.log
# fused dequantisation vs the resident fp16 panel, same kernel, by batch size
timeout 300 python scripts/gemm_bench.py q4_0 0 160128 128128 b32.qkv b32.up b32.out b32.down b32.b32.qkv b32.b32.up b32.b32.out b32.b32.down b64.up b64.down b128.up b128.down txt.qkv txt.up txt.out txt.down 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_bench.log
timeout 300 python scripts/gemm_bench.py q4_0 pre 0 160128 128128 b32.qkv b32.up b32.out b32.down b32.b32.qkv b32.b32.up b32.b32.out b32.b32.down b64.up b64.down b128.up b128.down txt.qkv txt.up txt.out txt.down 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_bench.log
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench.json
CLIP_AMD_RESIDENT_F16=0 timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench_noresident.json
timeout 600 python bench.py --config l14_f16_b256 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench_l14.json
