#!/bin/bash
# One gpurun call: GPU test tier + smoke + bench + rocprofv3 kernel trace. Logs under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r01}
echo "== kernels tests" ; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --maxfail=8 2>&1 | tail -40 | tee gpurun_out/${TAG}_kernels.log
echo "== parity tests" ; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=12 2>&1 | tail -60 | tee gpurun_out/${TAG}_parity.log
echo "== smoke" ; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee gpurun_out/${TAG}_smoke.log
echo "== bench" ; timeout 900 python bench.py --steps 10 --warmup 3 --json-out gpurun_out/${TAG}_bench.json 2>&1 | tail -5 | tee gpurun_out/${TAG}_bench.log
echo "== rocprof" ; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o ${TAG} -- python "${GRAFT_REPO_ROOT:-/root/repo}/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/prof_${TAG}.log 2>&1; tail -3 /tmp/prof_${TAG}.log)
find /tmp/prof_${TAG} -type f | head -20; for f in $(find /tmp/prof_${TAG} -name "*kernel_stats*.csv"); do cp $f gpurun_out/${TAG}_kernel_stats.csv; done
head -30 gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
