#!/bin/bash
# One gpurun call: GPU test tier + smoke + bench + rocprofv3 kernel trace (+ PMC HBM traffic). Logs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02}
echo "== kernels tests" ; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/${TAG}_kernels.log
echo "== parity tests" ; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -15 | tee gpurun_out/${TAG}_parity.log
echo "== smoke" ; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.log
echo "== bench" ; timeout 900 python bench.py --json-out gpurun_out/${TAG}_bench.json 2>&1 | tail -2 | cut -c1-3000 | tee gpurun_out/${TAG}_bench.log
echo "== rocprof kernel trace" ; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python "${GRAFT_REPO_ROOT:-/root/repo}/bench.py" --steps 10 --warmup 3 --preheat 0.3 --no-cpu-baseline --no-roofline > /tmp/prof_${TAG}.log 2>&1; tail -1 /tmp/prof_${TAG}.log | cut -c1-300)
for f in $(find /tmp/prof_${TAG} -name "*kernel_stats*.csv"); do grep -v "at::native\|__amd_rocclr" $f | cut -c1-400 > gpurun_out/${TAG}_kernel_stats.csv; done
head -8 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
echo "== rocprof PMC (HBM traffic)"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$n -o pmc -- python "${GRAFT_REPO_ROOT:-/root/repo}/bench.py" --steps 3 --warmup 1 --preheat 0 --no-cpu-baseline --no-roofline > /tmp/pmc_${TAG}_$n.log 2>&1)
done
python - <<PY | tee gpurun_out/${TAG}_pmc_traffic.txt
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "clipamd" not in k: continue
        k = k.replace("void clipamd::(anonymous namespace)::", "").split("(")[0].replace(" ", "")
        if k.startswith("_ZN7clipamd"): k = k.split("I")[0].replace("_ZN7clipamd12_GLOBAL__N_1", "")[2:] if False else k
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
out = {}
for k in acc:
    fs = acc[k].get("FETCH_SIZE", 0) / max(1, cnt[(k, "FETCH_SIZE")]); ws = acc[k].get("WRITE_SIZE", 0) / max(1, cnt[(k, "WRITE_SIZE")])
    # rocprofv3 units: KB per dispatch; gfx950 correction (MI355X_MICROARCH.md): FETCH_SIZE reads 1/2 of a wide coalesced stream -> x2
    out[k] = {"fetch_kb_raw": fs, "write_kb_raw": ws, "hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0, "launches": cnt[(k, "FETCH_SIZE")]}
    print("%-60s FETCH_SIZE %10.1f KB  WRITE_SIZE %10.1f KB  -> HBM bytes/launch (fetch x2) %.3e  [%d launches]" % (k[:60], fs, ws, out[k]["hbm_bytes_per_launch"], cnt[(k, "FETCH_SIZE")]))
import sys
sys.path.insert(0, ".")
import bench
out["_kernel_src_sha16"] = bench.kernel_source_sha16()      # bench.py reports traffic only while the kernel sources are these
out["_config"] = "b32_q4_0_b256"
json.dump(out, open("gpurun_out/${TAG}_pmc_traffic.json", "w"), indent=1)
print("-- MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES summed over all SIMDs; 16 cycles per v_mfma_f32_16x16x32_f16; GRBM_GUI_ACTIVE = GPU-active cycles of the dispatch)")
for k in acc:
    a = {c: acc[k][c] / max(1, cnt[(k, c)]) for c in acc[k]}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and a.get("GRBM_GUI_ACTIVE", 0) > 0:
        print("%-60s MFMA_BUSY %.3e  INSTS_MFMA %.3e  BUSY_CU %.3e  GUI_ACTIVE %.3e  -> MFMA-busy fraction = busy / (GUI_ACTIVE/8 XCDs x 1024 SIMDs) = %.3f" % (
            k[:60], a["SQ_VALU_MFMA_BUSY_CYCLES"], a.get("SQ_INSTS_MFMA", 0), a.get("SQ_BUSY_CU_CYCLES", 0), a["GRBM_GUI_ACTIVE"],
            a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)))
PY
