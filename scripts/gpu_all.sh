#!/bin/bash
# full GPU tier as the driver runs it + smoke
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu 2>&1 | tail -15
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
