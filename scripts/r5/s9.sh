#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05i
Q="--no-matrix --no-cpu-baseline --no-host-api --no-roofline"
for rep in 1 2 3; do for on in 0 1; do
  echo "== bench BENCH_EXP_PIPELINE=$on (run $rep)"; BENCH_EXP_PIPELINE=$on timeout 300 python bench.py $Q 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['images_per_s_per_gpu'], d['texts_per_s_per_gpu'])"
done; done 2>&1 | tee gpurun_out/${T}_pipeline_ab.txt
