#!/bin/bash
# mid-M ring kernel: tests + micro-benchmark against the two-buffer tiles on the batch-32 / text shapes (+ phase stamps with a tim variant)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-ring}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "ring or split_k" 2>&1 | tail -3 | tee gpurun_out/${TAG}_tests.log
GEMM_ITERS=200 timeout 300 python scripts/gemm_bench.py q4_0 0 65128 65256 2065256 4065256 66128 66256 2066256 3066256 4066256 b32.b32.qkv b32.b32.out b32.b32.up b32.b32.down b64.qkv b64.out b64.up b64.down 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_bench.log
GEMM_ITERS=200 timeout 300 python scripts/gemm_bench.py f16 0 65064 2065064 65128 66128 2066128 l14.b1.qkv l14.b1.out l14.b1.up l14.b1.down 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_bench.log
if [ -f clip_cpp_amd/variants/libclip_tim.so ]; then
  for cfg in "66256" "66128"; do
    rm -f /tmp/stamps.txt
    CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_tim.so CLIPAMD_G8_STAMPS=/tmp/stamps.txt GEMM_ITERS=100 timeout 300 python scripts/gemm_bench.py q4_0 $cfg b32.b32.up b32.b32.down 2>&1 | grep -v amdgpu.ids
    python scripts/g8_stamps.py /tmp/stamps.txt | grep -v "workgroup start"
  done 2>&1 | tee gpurun_out/${TAG}_stamps.txt
fi
