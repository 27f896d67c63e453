#!/bin/bash
# final validation of the host-side changes made after the s23 measurement set (image formats, loader guard): GPU tier + smoke, no timing
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
echo "== GPU tests" ; timeout 1500 python -m pytest tests/ -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E " passed| failed| error|FAILED|ERROR" | tail -8 | tee gpurun_out/r05w_tests.log
echo "== smoke" ; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2 | tee gpurun_out/r05w_smoke.log
echo "== image formats on this host" ; timeout 300 python -m pytest tests/test_image_io.py tests/test_malformed_inputs.py -q -p no:cacheprovider 2>&1 | tail -1 | tee -a gpurun_out/r05w_tests.log
