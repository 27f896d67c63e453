#!/bin/bash
# after the last kernel-source change: PMC HBM traffic at the final sources, the batch-1 configs, one more default bench line, GPU tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02z2}
R="${GRAFT_REPO_ROOT:-/root/repo}"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && timeout 80 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$set -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --preheat 0 --no-cpu-baseline --no-roofline --no-host-api > /tmp/pmc_${TAG}_$set.log 2>&1)
done
python - <<PY | tee gpurun_out/${TAG}_pmc_traffic.txt
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "clipamd" not in k: continue
        k = k.replace("void clipamd::(anonymous namespace)::", "").split("(")[0].replace(" ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
out = {}
for k in acc:
    fs = acc[k].get("FETCH_SIZE", 0) / max(1, cnt[(k, "FETCH_SIZE")]); ws = acc[k].get("WRITE_SIZE", 0) / max(1, cnt[(k, "WRITE_SIZE")])
    out[k] = {"fetch_kb_raw": fs, "write_kb_raw": ws, "hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0, "launches": cnt[(k, "FETCH_SIZE")]}
    print("%-60s FETCH_SIZE %10.1f KB  WRITE_SIZE %10.1f KB  -> HBM bytes/launch (fetch x2) %.3e  [%d launches]" % (k[:60], fs, ws, out[k]["hbm_bytes_per_launch"], cnt[(k, "FETCH_SIZE")]))
import bench
out["_kernel_src_sha16"] = bench.kernel_source_sha16()
out["_config"] = "b32_q4_0_b256"
json.dump(out, open("gpurun_out/${TAG}_pmc_traffic.json", "w"), indent=1)
PY
for cfg in b32_q4_0_b1 l14_f16_b1; do
  timeout 60 python bench.py --config $cfg --no-cpu-baseline --no-host-api --json-out gpurun_out/${TAG}_cfg_$cfg.json 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu'])" | tee -a gpurun_out/${TAG}_configs.log
done
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
timeout 60 python bench.py --no-cpu-baseline --no-host-api --json-out gpurun_out/${TAG}_bench_nocpu.json 2>&1 | tail -1 | cut -c1-400
timeout 130 python -m pytest tests/ -m gpu -q -x 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -4 | tee gpurun_out/${TAG}_tests.log
