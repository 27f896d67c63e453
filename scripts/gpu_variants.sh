#!/bin/bash
# A/B the GEMM micro-benchmark over library variants built by scripts/build_variant.sh: gpu_variants.sh "<bench args>" name1 name2 ...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ARGS=$1; shift
echo "== base"; python scripts/gemm_bench.py $ARGS 2>&1 | grep -v amdgpu.ids
for v in "$@"; do
  echo "== $v"; CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_$v.so python scripts/gemm_bench.py $ARGS 2>&1 | grep -v amdgpu.ids
done
