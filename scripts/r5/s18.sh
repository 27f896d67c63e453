#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05q
echo "== attention + parity tests"; timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -x -k "attention or head_sizes or 336px or two_tower or text_tower_parity or baseline_batch or configs_3_4_5 or every_batch_size" 2>&1 | grep -E " passed| failed|Error|assert" | tail -5 | tee gpurun_out/${T}_tests.log
Q="--no-matrix --no-cpu-baseline --no-host-api --no-rates"
for rep in 1 2; do
  echo "== default"; timeout 600 python bench.py $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); k=d['kernels']
for n,v in list(k.items()): 
    if n.startswith('attention'): print('   %-70s %8.4f ms/step x%d  %s TF' % (n, v['ms_per_step'], v['launches_per_step'], v['tflops']))"
done 2>&1 | tee gpurun_out/${T}_attn_bench.txt
echo "== cfg3"; timeout 600 python bench.py --config cfg3_l14_f16_b256_img $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); k=d['kernels']
for n,v in list(k.items()): 
    if n.startswith('attention'): print('   %-70s %8.4f ms/step x%d  %s TF' % (n, v['ms_per_step'], v['launches_per_step'], v['tflops']))" | tee -a gpurun_out/${T}_attn_bench.txt
echo "== cfg5"; timeout 600 python bench.py --config cfg5_h14_q8_0_b64_img $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); k=d['kernels']
for n,v in list(k.items()): 
    if n.startswith('attention'): print('   %-70s %8.4f ms/step x%d  %s TF' % (n, v['ms_per_step'], v['launches_per_step'], v['tflops']))" | tee -a gpurun_out/${T}_attn_bench.txt
