#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05r
export CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_attnabl.so
Q="--no-matrix --no-cpu-baseline --no-host-api --no-rates"
for dbg in 0 1 2 4 3 5 6; do
  echo -n "cfg3 attention debug=$dbg (1 no staging loads, 2 no query blocks, 4 no stores): "; CLIP_AMD_ATTN_DEBUG=$dbg timeout 600 python bench.py --config cfg3_l14_f16_b256_img $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
for n,v in list(k.items()): 
    if n.startswith('attention'): print('%s %8.4f ms/step x%d -> %.1f us per launch' % (n, v['ms_per_step'], v['launches_per_step'], v['ms_per_step']/v['launches_per_step']*1e3))"
done 2>&1 | tee gpurun_out/${T}_attn_ablation.txt
for dbg in 0 1 2 4; do
  echo -n "default attention debug=$dbg: "; CLIP_AMD_ATTN_DEBUG=$dbg timeout 600 python bench.py $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
print(' | '.join('%s %.1f us' % (n, v['ms_per_step']/v['launches_per_step']*1e3) for n,v in k.items() if n.startswith('attention')))"
done 2>&1 | tee -a gpurun_out/${T}_attn_ablation.txt
