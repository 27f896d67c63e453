#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05g
echo "== kernel tests (fold, gemm4, gemm8, gemm8p)"; timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -x -k "lnfold or gemm8 or large_m or gemm4" 2>&1 | tail -5 | tee gpurun_out/${T}_tests.log
echo "== parity tests touching the resident panels"; timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "resident or baseline_batch or every_dispatch or full_size" 2>&1 | tail -5 | tee -a gpurun_out/${T}_tests.log
Q="--no-matrix --no-cpu-baseline --no-host-api"
for rep in 1 2 3; do for on in 0 1; do
  echo "== bench CLIP_AMD_GEMM4_F16OUT=$on (run $rep)"; CLIP_AMD_GEMM4_F16OUT=$on timeout 300 python bench.py $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu']); k=d['kernels']
for n,v in list(k.items())[:10]: print('   %-70s %8.4f ms/step x%d  %s TF' % (n, v['ms_per_step'], v['launches_per_step'], v['tflops']))"
done; done 2>&1 | tee gpurun_out/${T}_bench_ab.txt
