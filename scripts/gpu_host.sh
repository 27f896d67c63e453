#!/bin/bash
# host-pointer API: forward granularity (CLIP_AMD_HOST_SUBCHUNK) x copy piece (CLIP_AMD_HOST_COPY_PIECE) x packer threads
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-host}
nproc
run() { timeout 200 python scripts/host_api_bench.py $1 2>&1 | grep -v amdgpu.ids | awk '/encode_images_from_host/ {last=$0; next} {if (last != "") print "   " last; last=""; print}'; }
for fg in 64 128 256; do echo "== forward group $fg, copy piece 32"; CLIP_AMD_HOST_SUBCHUNK=$fg CLIP_AMD_HOST_TIMING=1 run 256; done 2>&1 | tee gpurun_out/${TAG}_host_api.log
for fg in 128 256; do echo "== batch 1024, forward group $fg"; CLIP_AMD_HOST_SUBCHUNK=$fg CLIP_AMD_HOST_TIMING=1 run 1024; done | tee -a gpurun_out/${TAG}_host_api.log
echo "== batch 64 (defaults)"; run 64 | tee -a gpurun_out/${TAG}_host_api.log
echo "== batch 16 (defaults)"; run 16 | tee -a gpurun_out/${TAG}_host_api.log
