#!/bin/bash
# Register / LDS / scratch usage of the kernels in one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
#   scripts/kernel_resources.sh k_gemm.hip -DCLIPAMD_GEMM_WT=1 [filter-regex]
cd "$(dirname "$0")/.." || exit 1
SRC=$1; DEF=${2:-}; FILT=${3:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-inline-asm -Iinclude $DEF \
  -Rpass-analysis=kernel-resource-usage -c clip_cpp_amd/csrc/$SRC -o /tmp/kres.o 2>&1 | grep "remark:" | \
python3 -c '
import re, sys, subprocess
cur = {}; rows = []
for line in sys.stdin:
    t = line.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
    k, _, v = t.partition(":")
    if k.strip() == "Function Name":
        cur = {"name": v.strip()}; rows.append(cur)
    else:
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("void clipamd::(anonymous namespace)::", "").split("(")[0]
    print("%-56s v=%-4s a=%-4s s=%-4s scratch=%-5s occ=%-2s spillV=%s" % (name[:56], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill")))
' | grep -E "$FILT"
