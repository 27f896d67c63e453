#!/bin/bash
# what MFMA rate does the part sustain?  the 4-wave kernel with its LDS-DMA requests and fragment reads compiled out (MFMAs + barriers only),
# long K, many launches; core clock from s_memtime / s_memrealtime stamps, power and sclk from rocm-smi
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-pure}
( for i in $(seq 1 120); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed 's/.*: *//' | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/${TAG}_smi.log 2>&1 &
SMI=$!
for v in g4PURE ""; do
  if [ -n "$v" ]; then export CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_$v.so; else unset CLIP_AMD_LIB; fi
  for it in 20 3000; do
    GEMM_ITERS=$it timeout 300 python scripts/gemm_bench.py f16 256259 sq.k8k sq8.k4k 2>&1 | grep -v amdgpu.ids | sed "s/^/${v:-full} iters=$it /" | tee -a gpurun_out/${TAG}.log
  done
done
kill $SMI 2>/dev/null
for v in g4PUREtim g4tim; do
  rm -f /tmp/stamps.txt
  CLIPAMD_G8_STAMPS=/tmp/stamps.txt CLIP_AMD_LIB=$PWD/clip_cpp_amd/variants/libclip_$v.so GEMM_ITERS=500 timeout 300 python scripts/gemm_bench.py f16 256259 sq.k8k 2>&1 | grep -v amdgpu.ids
  echo "== $v" | tee -a gpurun_out/${TAG}.log; python scripts/g8_stamps.py /tmp/stamps.txt | tee -a gpurun_out/${TAG}.log
done
awk '{w=$NF+0; c=$1; gsub(/[^0-9]/,"",c); if (w>600) print c, w}' gpurun_out/${TAG}_smi.log | sort -n | awk '{a[NR]=$0} END {print "busy smi samples (sclk MHz, W): first", a[1], "| median", a[int(NR/2)+1], "| last", a[NR], "| n", NR}' | tee -a gpurun_out/${TAG}.log
