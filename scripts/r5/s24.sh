#!/bin/bash
# throughput of the f32-file path (k_gemm_f32.hip) beside the f16 and q4_0 files of the same model, BASELINE step shape (256 images + 256 texts)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for ft in f32 f16 q4_0; do
  st=100; [ $ft = f32 ] && st=10
  echo "== b32 $ft"
  timeout 600 python bench.py --ftype $ft --steps $st --warmup 2 --no-matrix --no-cpu-baseline --no-roofline --no-host-api 2>&1 | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('%s: %.1f emb/s  %.3f ms/step  img %.1f/s  txt %.1f/s' % ('$ft', d['value'], d['ms_per_step'], d.get('images_per_s_per_gpu', 0), d.get('texts_per_s_per_gpu', 0)))"
done 2>&1 | tee gpurun_out/r05v_f32_throughput.txt
