#!/bin/bash
# mid-M path after wiring the ring kernel into pick_tile: GPU test tier (kernels + parity) and the batch-2..64 / text-batch bench lines
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-mid}
timeout 1500 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/${TAG}_tests.log
for cfg in cfg2_b32_q4_0_b32_img b32_q4_0_b32 l14_f16_b1 b32_q4_0_b1; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-host-api --json-out gpurun_out/${TAG}_$cfg.json 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu']); [print('   ', k, v) for k, v in (d.get('kernels') or {}).items()]" | tee -a gpurun_out/${TAG}_bench.log
done
for b in 2 4 8 16 64; do
  timeout 300 python bench.py --config b32_q4_0_b1 --batch $b --no-cpu-baseline --no-roofline --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('b32 q4_0 batch $b (+ $b texts)', d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu'])" | tee -a gpurun_out/${TAG}_bench.log
done
CLIP_AMD_SKINNY_ROWS=128 timeout 300 python bench.py --config b32_q4_0_b1 --batch 2 --no-cpu-baseline --no-roofline --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('b32 q4_0 batch 2 skinny<=128', d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu'])" | tee -a gpurun_out/${TAG}_bench.log
