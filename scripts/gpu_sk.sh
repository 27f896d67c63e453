#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-sk}
echo "== kernel tests (skinny)"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k skinny 2>&1 | tail -12
echo "== parity + multi tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_device.py -m gpu -q 2>&1 | tail -12
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
for cfg in b32_q4_0_b1 b32_q4_0_b32 l14_f16_b1; do
  echo "== bench $cfg (skinny on)"; timeout 300 python bench.py --config $cfg --vision-only --no-cpu-baseline --no-host-api --json-out gpurun_out/${TAG}_$cfg.json 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['whole_step_roofline']); [print('   ',k,v) for k,v in (d['kernels'] or {}).items()]"
done
echo "== bench b32_q4_0_b1 (skinny off)"; CLIP_AMD_SKINNY=0 timeout 300 python bench.py --config b32_q4_0_b1 --vision-only --no-cpu-baseline --no-host-api --no-roofline 2>&1 | tail -1 | cut -c1-200
echo "== latency"; timeout 200 python scripts/latency.py 2>&1 | tail -6
