#!/bin/bash
# Which hipBLASLt/rocBLAS kernels (macro-tile, wave grid, ...) the vendor library picks for our GEMM shapes: kernel names from rocprofv3.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/blasn -o blasn -- python "${GRAFT_REPO_ROOT:-/root/repo}/scripts/gemm_bench.py" blas > /tmp/blasn.log 2>&1)
grep -v amdgpu.ids /tmp/blasn.log | tail -20
for f in $(find /tmp/blasn -name "*kernel_stats*.csv"); do cut -c1-700 $f | grep -i "Cijk\|gemm" > gpurun_out/blas_kernel_names.csv; done
cat gpurun_out/blas_kernel_names.csv
