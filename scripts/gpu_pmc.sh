#!/bin/bash
# PMC counters for the GEMM microbench: usage gpu_pmc.sh TAG "<gemm_bench args>"
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-pmc}; ARGS=${2:-"128128 q4_0 b32.down b32.up"}
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -2
echo "== gemm bench"; timeout 600 python scripts/gemm_bench.py $ARGS 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee gpurun_out/${TAG}_gemm_bench.log
: > gpurun_out/${TAG}_pmc.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$n -o pmc -- python "${GRAFT_REPO_ROOT:-/root/repo}/scripts/gemm_bench.py" $ARGS > /tmp/pmc_$n.log 2>&1; grep -iE "error|invalid" /tmp/pmc_$n.log | head -3)
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY' | tee -a gpurun_out/${TAG}_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "clipamd" not in k: continue
    k = k.replace("void clipamd::(anonymous namespace)::", "").split("(")[0] + " grid=" + r.get("Grid_Size", "?")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in acc[k].items(): print("   %-28s %16.1f  (avg of %d)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
  fi
done
