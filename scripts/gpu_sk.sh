#!/bin/bash
# small-M path: kernel tests + end-to-end parity of the batch-1 / single-text paths + batch-1 bench lines (row limit sweep)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-sk}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "skinny" 2>&1 | tail -3 | tee gpurun_out/${TAG}_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "small_models or vit_b32_q4_0_batch_parity or graph_replay or text_tower_parity" 2>&1 | tail -3 | tee -a gpurun_out/${TAG}_tests.log
for cfg in b32_q4_0_b1 l14_f16_b1; do
  timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-roofline --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu'])" | tee -a gpurun_out/${TAG}_bench.log
done
for rows in 128 256 512; do
 for b in 2 4 8; do
  CLIP_AMD_SKINNY_ROWS=$rows timeout 300 python bench.py --config b32_q4_0_b1 --batch $b --texts 0 --no-cpu-baseline --no-roofline --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rows<=$rows batch $b', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${TAG}_bench.log
 done
done
