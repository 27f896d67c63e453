#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05s
export CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_attnabl.so
Q="--no-matrix --no-cpu-baseline --no-host-api --no-rates"
for dl in 0 4 8 12 16 24 32; do
  echo -n "cfg3 second-slot workgroups delayed by $dl x 3.9 us: "; CLIP_AMD_ATTN_DEBUG=$((dl*16)) timeout 600 python bench.py --config cfg3_l14_f16_b256_img $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']
for n,v in list(k.items()): 
    if n.startswith('attention'): print('%.1f us per launch' % (v['ms_per_step']/v['launches_per_step']*1e3))"
done 2>&1 | tee gpurun_out/${T}_attn_dephase.txt
