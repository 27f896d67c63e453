#!/bin/bash
# the GPU-tier tests that decode image files (the reference's example programs compiled unchanged, the host API), after the decoder changes; + smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_host_api.py -m gpu -q -p no:cacheprovider -k "reference or example or program or zsl or extract or benchmark or main or image_file or load" 2>&1 | tail -3 | tee gpurun_out/r05y_image_callers.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1 | tee -a gpurun_out/r05y_image_callers.log
