#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-host}
nproc
for sc in 32 64 128 256; do echo "== subchunk $sc"; CLIP_AMD_HOST_SUBCHUNK=$sc CLIP_AMD_HOST_TIMING=1 timeout 200 python scripts/host_api_bench.py 256 2>&1 | grep -v amdgpu.ids | awk '/encode_images_from_host/ {last=$0; next} {if (last != "") print "   " last; last=""; print}' ; done 2>&1 | tee gpurun_out/${TAG}_host_api.log
echo "== batch 1024"; CLIP_AMD_HOST_TIMING=1 timeout 200 python scripts/host_api_bench.py 1024 2>&1 | grep -v amdgpu.ids | awk '/encode_images_from_host/ {last=$0; next} {if (last != "") print "   " last; last=""; print}' | tee -a gpurun_out/${TAG}_host_api.log
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q 2>&1 | tail -3
echo "== parity tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py -m gpu -q 2>&1 | tail -5
