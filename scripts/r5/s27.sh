#!/bin/bash
# flake hunt: the bench contract test 30 times on one box, every failure's assertion kept
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; : > gpurun_out/r05x_contract_repeat.txt
for i in $(seq 1 30); do
  timeout 300 python -m pytest tests/test_bench_contract.py -m gpu -q -x -p no:cacheprovider -k contract_fields 2>&1 | grep -E "^E  |passed|failed" | head -12 | sed "s/^/run $i: /" >> gpurun_out/r05x_contract_repeat.txt
done
grep -c "1 passed" gpurun_out/r05x_contract_repeat.txt
