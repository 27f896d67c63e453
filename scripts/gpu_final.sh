#!/bin/bash
# lean end-of-session measurement set (GPU budget): bench (default config, full JSON) -> GPU test tier -> rocprofv3 kernel trace ->
# PMC HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) -> the small / mid batch configs.  Logs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02y}
R="${GRAFT_REPO_ROOT:-/root/repo}"
echo "== bench" ; timeout 240 python bench.py --json-out gpurun_out/${TAG}_bench.json 2>&1 | tail -1 | cut -c1-600
echo "== GPU tests" ; timeout 200 python -m pytest tests/ -m gpu -q -x 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -6 | tee gpurun_out/${TAG}_tests.log
echo "== rocprof kernel trace (default config)" ; (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python "$R/bench.py" --steps 10 --warmup 3 --preheat 0.3 --no-cpu-baseline --no-roofline --no-host-api > /tmp/prof_${TAG}.log 2>&1; tail -1 /tmp/prof_${TAG}.log | cut -c1-200)
for f in $(find /tmp/prof_${TAG} -name "*kernel_stats*.csv"); do grep -v "at::native\|__amd_rocclr" $f | cut -c1-400 > gpurun_out/${TAG}_kernel_stats_b32_q4_0_b256.csv; done
head -6 gpurun_out/${TAG}_kernel_stats_b32_q4_0_b256.csv | cut -c1-160
echo "== rocprof PMC (HBM traffic)"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$n -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --preheat 0 --no-cpu-baseline --no-roofline --no-host-api > /tmp/pmc_${TAG}_$n.log 2>&1)
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
echo "== small / mid batch configs"
for cfg in b32_q4_0_b1 cfg2_b32_q4_0_b32_img b32_q4_0_b32 l14_f16_b1 l14_f16_b32; do
  timeout 60 python bench.py --config $cfg --no-cpu-baseline --no-host-api --json-out gpurun_out/${TAG}_cfg_$cfg.json 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu'])" | tee -a gpurun_out/${TAG}_configs.log
done
