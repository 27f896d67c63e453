#!/bin/bash
# Round 3, late: (1) parity subset for the 4-element text-embedding dequantiser, (2) rocprofv3 kernel stats of config 2 (32 images, the
# CPU-runnable case of BASELINE.json), (3) the default bench line without matrix / CPU leg.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
echo "== tests (text)"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "text or tiny or ftype" 2>&1 | tail -3
echo "== cfg2 kernel stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg2 -o cfg2 -- python "$R/bench.py" --config cfg2_b32_q4_0_b32_img --steps 200 --warmup 20 --preheat 0.3 --no-cpu-baseline --no-roofline --no-host-api > /tmp/prof_cfg2.log 2>&1; tail -1 /tmp/prof_cfg2.log | cut -c1-200)
for f in $(find /tmp/prof_cfg2 -name "*kernel_stats*.csv"); do grep -v "at::native\|__amd_rocclr" $f | cut -c1-400 > gpurun_out/r03_kernel_stats_cfg2.csv; done
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03_kernel_stats_cfg2.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print('%-100s calls %6s avg %8.2f us %5.2f%%'%(r['Name'][:100], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
echo "== bench (default, no matrix)"; timeout 600 python bench.py --no-matrix --no-cpu-baseline --no-host-api 2>&1 | tail -1 | cut -c1-400
echo "== bench cfg2"; timeout 600 python bench.py --config cfg2_b32_q4_0_b32_img --no-cpu-baseline --no-host-api --no-roofline 2>&1 | tail -1 | cut -c1-300
