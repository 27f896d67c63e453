#!/bin/bash
# GEMM experiment round: correctness of all tile variants, micro-benchmark, PMC counters.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-exp}
echo "== kernel tests"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -5
echo "== gemm bench"; timeout 600 python scripts/gemm_bench.py 128128 1128128 64128 1064128 128064 64064 q4_0 2>&1 | tee gpurun_out/${TAG}_gemm_bench.log
echo "== f16/q8"; timeout 600 python scripts/gemm_bench.py 128128 1128128 f16 q8_0 b32.up b32.down 2>&1 | tee -a gpurun_out/${TAG}_gemm_bench.log
(cd /tmp && rocprofv3 -L > /tmp/counters.txt 2>&1; grep -oE "^\s*(Name|Counter_Name)\s*:\s*\S+|\b(SQ|TCC|TCP|GRBM|TA)_[A-Z0-9_]+\b" /tmp/counters.txt | sort -u | head -400 > "${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/${TAG}_counters.txt"; wc -l /tmp/counters.txt)
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$n -o pmc -- python "${GRAFT_REPO_ROOT:-/root/repo}/scripts/gemm_bench.py" 128128 1128128 q4_0 b32.down b32.up > /tmp/pmc_$n.log 2>&1; tail -2 /tmp/pmc_$n.log)
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY' | tee -a gpurun_out/${TAG}_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")[:100]
    if "gemm_kernel" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in acc[k].items(): print("   %-28s %16.1f  (per dispatch over %d)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
  fi
done
