#!/bin/bash
# kernel-trace of the batch-1 / batch-32 vision forward: per-kernel durations vs wall time (launch gaps)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for B in 1 32; do
  python bench.py --model b32 --ftype q4_0 --batch $B --vision-only --steps 50 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b1prof$B -o p -- python "${GRAFT_REPO_ROOT:-/root/repo}/bench.py" --model b32 --ftype q4_0 --batch $B --vision-only --steps 50 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/b1prof$B.log 2>&1)
  for f in $(find /tmp/b1prof$B -name "*kernel_stats*.csv"); do grep -v "at::native\|__amd_rocclr" $f | cut -c1-260 > gpurun_out/b${B}_kernel_stats.csv; done
  python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/b${B}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("batch $B: sum of kernel time per step (53 steps): %.1f us" % (tot/53/1e3))
for r in rows[:14]: print("  %-100s calls %5s avg %8.1f us  %5s%%" % (r["Name"].replace("void clipamd::(anonymous namespace)::","")[:100], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
