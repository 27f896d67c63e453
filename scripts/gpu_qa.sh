#!/bin/bash
# fused LN1 + q/k/v + attention kernel of the small-M path: A/B + oracle test, graph tests, batch-1 bench lines with and without it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-qa}
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fused_ln_qkv or graph_replay or small_models" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee gpurun_out/${TAG}_tests.log
for f in 1 0; do
  for cfg in b32_q4_0_b1 l14_f16_b1; do
    CLIP_AMD_FUSED_QKV=$f timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused=$f $cfg', d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu']); [print('   ', k, v) for k, v in list((d.get('kernels') or {}).items())[:12]]" | tee -a gpurun_out/${TAG}_bench.log
  done
done
