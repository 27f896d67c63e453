#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-tim}
rm -f /tmp/stamps.txt
CLIPAMD_G8_STAMPS=/tmp/stamps.txt CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_tim.so timeout 300 python scripts/gemm_bench.py f16 160256 b32.qkv b32.out b32.up b32.down 2>&1 | grep -v amdgpu.ids
python scripts/g8_stamps.py /tmp/stamps.txt | tee gpurun_out/${TAG}_stamps.txt
