#!/bin/bash
# epilogue rework (batched bias, residual strips one ahead, counted waits): correctness, then A/B on all kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-epi}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu 2>&1 | tail -4 | tee gpurun_out/${TAG}_tests.log
GEMM_ITERS=300 timeout 400 python scripts/gemm_bench.py q4_0 0 160128 b32.qkv b32.up b32.out b32.down txt.qkv txt.up txt.out txt.down 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_bench.log
GEMM_ITERS=200 timeout 400 python scripts/gemm_bench.py f16 0 160128 160256 256258 256260 l14.up l14.down l14.qkv l14.out sq.k1k 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_bench.log
timeout 300 python bench.py --no-cpu-baseline --no-host-api 2>&1 | tail -1 > gpurun_out/${TAG}_bench.json; python - <<PY
import json; d=json.load(open("gpurun_out/${TAG}_bench.json")); print("bench value %.1f ms/step %.4f roofline %s %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel"][:30], d["roofline"]["frac"]))
for k,v in d["kernels"].items(): print("   %-60s %.4f ms %s" % (k, v["ms_per_step"], v["tflops"]))
PY
