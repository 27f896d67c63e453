#!/bin/bash
# both GPU test tiers + the default and ViT-L/14 bench lines (no profiles)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-chk}
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 | tee gpurun_out/${TAG}_tests.log
for cfg in b32_q4_0_b256 l14_f16_b256; do
  timeout 400 python bench.py --config $cfg --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/${TAG}_bench_$cfg.json
  python - <<PY
import json; d=json.load(open("gpurun_out/${TAG}_bench_$cfg.json")); r=d["roofline"]
print("$cfg: value %.1f ms/step %.4f img/s %.1f host_api %s | roofline %s frac %.3f" % (d["value"], d["ms_per_step"], d["images_per_s_per_gpu"], d.get("host_api_images_per_s"), r["kernel"][:34], r["frac"]))
for k,v in list(d["kernels"].items())[:8]: print("   %-62s %.4f ms %s" % (k, v["ms_per_step"], v["tflops"]))
PY
done 2>&1 | tee gpurun_out/${TAG}_bench.log
