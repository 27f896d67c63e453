#!/bin/bash
# Round 4, call A: the new tests (LayerNorm-fold centring, two-tower multi call, bench contract), the K-loop-only evidence for the
# fused-FFN decision (64-row tiles at the FFN shapes, epilogue compiled out), the default bench line and the single-process form.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r04_lnfold_centre_sweep.txt gpurun_out/r04_lnfold_centre_e2e.txt
echo "== new / changed GPU tests"
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "lnfold" 2>&1 | tail -5 | tee gpurun_out/r04a_tests_lnfold.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fold or common_mode or config2 or text_tower" 2>&1 | tail -8 | tee gpurun_out/r04a_tests_parity.log
timeout 900 python -m pytest tests/test_multi_device.py tests/test_bench_contract.py -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r04a_tests_multi_bench.log
echo "== 64-row tiles at the FFN shapes: whole kernel (dbg0) and K loop alone (dbg8), ablation build"
CLIP_AMD_LIB=clip_cpp_amd/variants/libclip_abl.so GEMM_ITERS=200 timeout 600 python scripts/gemm_bench.py b32.up b32.down txt.up txt.down 64128 65128 128128 160128 dbg0 dbg8 2>&1 | tee gpurun_out/r04a_ffn_bm64_kloop.txt
echo "== bench (default line)"
timeout 900 python bench.py --json-out gpurun_out/r04a_bench.json 2>&1 | tail -1 | cut -c1-6000 | tee gpurun_out/r04a_bench.log
echo "== bench --single-process (one replica)"
timeout 600 python bench.py --gpus 1 --single-process --json-out gpurun_out/r04a_bench_sp.json 2>&1 | tail -1 | cut -c1-1500
