#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-p}
python - <<'PY'
import ctypes as C, numpy as np, sys
sys.path.insert(0, ".")
import torch
import clip_cpp_amd
from oracle import ref
L = clip_cpp_amd.lib()
def fp(a): return a.ctypes.data_as(C.POINTER(C.c_float))
def run(tid, raw, N, K, X, bias, resid, epi, tile):
    M = X.shape[0]; y = np.full((M, N), np.nan, np.float32)
    rc = L.clip_amd_test_gemm(tid, raw.ctypes.data_as(C.c_void_p), N, K, fp(X), M, fp(bias), fp(resid), fp(y), epi, tile)
    assert rc == 0, rc
    return y
rng = np.random.default_rng(1)
bad = 0
for (M, N, K) in [(333, 576, 192), (4000, 1024, 768), (12800, 2304, 768), (2051, 768, 64), (5000, 512, 128), (3000, 300, 1024)]:
    for tname, epi in [("f16", 0), ("q4_0", 1), ("q8_0", 3), ("f16", 4), ("q5_1", 2)]:
        tid = ref.GGML_TYPES[tname]
        W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        raw = ref.quantize(tid, W)
        X = rng.standard_normal((M, K)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32); resid = rng.standard_normal((M, N)).astype(np.float32)
        base = run(tid, raw, N, K, X, bias, resid, epi, 128128)
        for tile in (128257, 160257):
            for rep in range(2):
                y = run(tid, raw, N, K, X, bias, resid, epi, tile)
                if not np.array_equal(y, base):
                    bad += 1
                    print("MISMATCH", (M, N, K), tname, epi, tile, "max abs diff", np.nanmax(np.abs(y - base)), "nan", int(np.isnan(y).sum()))
print("persistent gemm8 bitwise check:", "OK" if not bad else "%d mismatches" % bad)
PY
timeout 300 python scripts/gemm_bench.py f16 160128 160256 128256 128257 160257 b32.qkv b32.up b32.out b32.down l14.up l14.down l14.qkv l14.out txt.qkv txt.up 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_persist.log
