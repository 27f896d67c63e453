mkdir -p gpurun_out
echo "== GPU tests"; timeout 2700 python -m pytest tests/ -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15 | tee gpurun_out/r06_tests_a.log
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3
echo "== bench"; timeout 900 python bench.py --json-out gpurun_out/r06_bench_a.json 2>&1 | tail -1 | cut -c1-1500
