#!/bin/bash
# ablation of the 4-wave kernel's K loop: which stream bounds it?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-g4abl}
SH="sq.k1k sq.k8k sq8.k4k l14.up l14.down"
for v in "" g4NODMA g4NOMFMA g4NOREAD; do
  if [ -n "$v" ]; then export CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_$v.so; else unset CLIP_AMD_LIB; fi
  GEMM_ITERS=200 timeout 300 python scripts/gemm_bench.py f16 256256 256259 $SH 2>&1 | grep -v amdgpu.ids | sed "s/^/${v:-full} /" | tee -a gpurun_out/${TAG}.log
done
