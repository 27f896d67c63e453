#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== bench (default, traffic from profiles/pmc_traffic.json)"; timeout 900 python bench.py --json-out gpurun_out/r05_bench.json 2>&1 | tail -1 | cut -c1-600
echo "== bench, driver form"; timeout 900 python bench.py --steps 20 --warmup 5 --json-out gpurun_out/r05_bench_driver_form.json 2>&1 | tail -1 | cut -c1-300
bash scripts/gpu_configs.sh r05cfg
