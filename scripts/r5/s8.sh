#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05h
echo "== f32 kernel + parity tests"; timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -x -k "f32" -s 2>&1 | grep -v "^$" | tail -12 | tee gpurun_out/${T}_tests.log
echo "== whole GPU tier"; timeout 1800 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -6 | tee -a gpurun_out/${T}_tests.log
