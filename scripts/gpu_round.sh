#!/bin/bash
# One gpurun call: GPU test tier + smoke + bench + rocprofv3 kernel trace (+ PMC HBM traffic / MFMA busy). Logs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r02}
R="${GRAFT_REPO_ROOT:-/root/repo}"
if [ "${PROFILE_ONLY:-0}" != "1" ]; then
echo "== GPU tests" ; timeout 2400 python -m pytest tests/ -m gpu -q -p no:cacheprovider 2>&1 | grep -E " passed| failed| error|FAILED|ERROR|max .* mean|1-cos" | tail -24 | tee gpurun_out/${TAG}_tests.log
echo "== smoke" ; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.log
echo "== bench" ; timeout 900 python bench.py --json-out gpurun_out/${TAG}_bench.json 2>&1 | tail -2 | cut -c1-4000 | tee gpurun_out/${TAG}_bench.log
fi
echo "== rocprof kernel trace (default config)" ; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python "$R/bench.py" --steps 10 --warmup 3 --preheat 0.3 --no-matrix --no-cpu-baseline --no-roofline --no-host-api > /tmp/prof_${TAG}.log 2>&1; tail -1 /tmp/prof_${TAG}.log | cut -c1-300)
for f in $(find /tmp/prof_${TAG} -name "*kernel_stats*.csv"); do grep -v "at::native\|__amd_rocclr" $f | cut -c1-400 > gpurun_out/${TAG}_kernel_stats_b32_q4_0_b256.csv; done
head -8 gpurun_out/${TAG}_kernel_stats_b32_q4_0_b256.csv | cut -c1-200
# the same trace split by (kernel name, grid): one instantiation serves several GEMM shapes (both towers), a per-name average mixes them
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/prof_${TAG}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "clipamd" not in k: continue
        gx = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0); wx = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 1)
        gy = int(r.get("Grid_Size_Y") or 1); wy = int(r.get("Workgroup_Size_Y") or 1)
        acc[(k.replace("void clipamd::(anonymous namespace)::", "").split("(")[0], (gx // max(1, wx)) * max(1, gy // max(1, wy)))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = sorted(((sum(v), k, v) for k, v in acc.items()), reverse=True)
with open("gpurun_out/${TAG}_kernel_stats_by_grid_b32_q4_0_b256.csv", "w") as o:
    o.write('"Name","Workgroups","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"\n')
    for tot, (name, wgs), v in rows:
        o.write('"%s",%d,%d,%d,%.1f,%d,%d\n' % (name, wgs, len(v), tot, tot / len(v), min(v), max(v)))
for tot, (name, wgs), v in rows[:10]: print("%-48s grid %5d WGs  x%5d  avg %8.1f us" % (name[:48], wgs, len(v), tot / len(v) / 1e3))
PY
echo "== rocprof kernel trace (batch 1, L/14 f16 batch 256)"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_b1 -o b1 -- python "$R/bench.py" --config b32_q4_0_b1 --vision-only --steps 50 --no-cpu-baseline --no-roofline --no-host-api > /tmp/prof_b1.log 2>&1)
for f in $(find /tmp/prof_${TAG}_b1 -name "*kernel_stats*.csv"); do grep -v "at::native\|__amd_rocclr" $f | cut -c1-400 > gpurun_out/${TAG}_kernel_stats_b32_q4_0_b1.csv; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_l14 -o l14 -- python "$R/bench.py" --config cfg3_l14_f16_b256_img --steps 3 --warmup 1 --preheat 0.3 --no-cpu-baseline --no-roofline --no-host-api > /tmp/prof_l14.log 2>&1)
for f in $(find /tmp/prof_${TAG}_l14 -name "*kernel_stats*.csv"); do grep -v "at::native\|__amd_rocclr" $f | cut -c1-400 > gpurun_out/${TAG}_kernel_stats_l14_f16_b256.csv; done
head -6 gpurun_out/${TAG}_kernel_stats_b32_q4_0_b1.csv | cut -c1-160; head -6 gpurun_out/${TAG}_kernel_stats_l14_f16_b256.csv | cut -c1-160
echo "== launch-boundary micro-benchmark"; [ -x scripts/ubench/launch_chain ] && timeout 300 ./scripts/ubench/launch_chain | tee gpurun_out/${TAG}_launch_chain.txt | tail -18
echo "== rocprof PMC (HBM traffic, MFMA busy)"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$n -o pmc -- python "$R/bench.py" --steps 3 --warmup 1 --preheat 0 --no-matrix --no-cpu-baseline --no-roofline --no-host-api --no-rates > /tmp/pmc_${TAG}_$n.log 2>&1)
done
python - <<PY | tee gpurun_out/${TAG}_pmc_traffic_and_mfma_busy.txt
import csv, glob, collections, json, sys
sys.path.insert(0, "$R")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "clipamd" not in k: continue
        k = k.replace("void clipamd::(anonymous namespace)::", "").split("(")[0].replace(" ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        try:      # ... and per (kernel name, workgroups of the dispatch): key "name@wgs" (bench.py roofline.grid_workgroups)
            kg = "%s@%d" % (k, int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
            acc[kg][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(kg, r["Counter_Name"])] += 1
        except (KeyError, ValueError):
            pass
out = {}
for k in acc:
    fs = acc[k].get("FETCH_SIZE", 0) / max(1, cnt[(k, "FETCH_SIZE")]); ws = acc[k].get("WRITE_SIZE", 0) / max(1, cnt[(k, "WRITE_SIZE")])
    # rocprofv3 units: KB per dispatch; gfx950 correction (MI355X_MICROARCH.md): FETCH_SIZE reads 1/2 of a wide coalesced stream -> x2
    out[k] = {"fetch_kb_raw": fs, "write_kb_raw": ws, "hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0, "launches": cnt[(k, "FETCH_SIZE")]}
    print("%-60s FETCH_SIZE %10.1f KB  WRITE_SIZE %10.1f KB  -> HBM bytes/launch (fetch x2) %.3e  [%d launches]" % (k[:60], fs, ws, out[k]["hbm_bytes_per_launch"], cnt[(k, "FETCH_SIZE")]))
import bench
# whole step: every launch of the PMC run belongs to one of its (1 warm-up + 3 timed) steps (--no-rates --preheat 0), except the load-time fold_kernel
tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in out.items() if "fold_kernel" not in k and "@" not in k)
out["_whole_step_hbm_bytes"] = tot / 4.0
print("whole step: %.3e HBM bytes (sum over every launch of a step; 4 steps in the PMC run)" % out["_whole_step_hbm_bytes"])
out["_kernel_src_sha16"] = bench.kernel_source_sha16()      # bench.py reports roofline.traffic only while the kernel sources are these
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
