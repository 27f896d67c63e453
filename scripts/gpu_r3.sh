#!/bin/bash
# Round-3 iteration runner (one gpurun call): GPU tests (optionally a -k filter), bench lines for the default config and config 2.
#   scripts/gpu_r3.sh TAG ["-k expr" | full | none] [extra bench configs ...]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r03}; shift
TESTS=${1:-full}; shift || true
if [ "$TESTS" = "full" ]; then
  timeout 1500 python -m pytest tests/ -m gpu -q 2>&1 | tail -60 | tee gpurun_out/${TAG}_tests.log
elif [ "$TESTS" != "none" ]; then
  timeout 1500 python -m pytest tests/ -m gpu -q $TESTS 2>&1 | tail -40 | tee gpurun_out/${TAG}_tests.log
fi
echo "== bench default"; timeout 600 python bench.py --json-out gpurun_out/${TAG}_bench.json 2>&1 | tail -1 | cut -c1-600
for cfg in "$@"; do
  echo "== bench $cfg"; timeout 600 python bench.py --config $cfg --no-host-api --json-out gpurun_out/${TAG}_cfg_${cfg}.json 2>&1 | tail -1 | cut -c1-400
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, e); continue
    print("==", f, d.get("value"), d.get("unit"), "ms/step", d.get("ms_per_step"), "cos", (d.get("cpu_baseline") or {}).get("gpu_vs_cpu_1_minus_cos_max"), (d.get("cpu_baseline") or {}).get("gpu_vs_cpu_text_1_minus_cos_max"))
    for k, v in (d.get("kernels") or {}).items(): print("   %-70s %8.4f ms  x%-3d %s" % (k, v["ms_per_step"], v["launches_per_step"], v["tflops"]))
PY
