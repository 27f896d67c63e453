#!/bin/bash
# per-workgroup phase stamps of a GEMM kernel (variant built with -DCLIPAMD_G8_TIMING): usage gpu_tim.sh TAG VARIANT TILE shapes...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-tim}; VAR=${2:-tim}; TILE=${3:-160256}; shift 3 || true
SH=${*:-b32.qkv b32.out b32.up b32.down}
rm -f /tmp/stamps.txt
CLIPAMD_G8_STAMPS=/tmp/stamps.txt CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_$VAR.so timeout 300 python scripts/gemm_bench.py f16 $TILE $SH 2>&1 | grep -v amdgpu.ids
python scripts/g8_stamps.py /tmp/stamps.txt | tee gpurun_out/${TAG}_stamps.txt
