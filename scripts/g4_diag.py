"""Diagnostic: the 4-wave 256x256 kernel (tile 256259) against the 64x64 tile, repeated; where do mismatches sit?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import clip_cpp_amd
from oracle import ref
L = clip_cpp_amd.lib()
def fp(a): return a.ctypes.data_as(C.POINTER(C.c_float))
def run(tid, raw, N, K, X, bias, resid, epi, tile):
    M = X.shape[0]; y = np.full((M, N), np.nan, np.float32)
    rc = L.clip_amd_test_gemm(tid, raw.ctypes.data_as(C.c_void_p), N, K, fp(X), M, fp(bias) if bias is not None else None,
                              fp(resid) if resid is not None else None, fp(y), epi, tile)
    assert rc == 0, rc
    return y
rng = np.random.default_rng(3)
TILE = int(os.environ.get("G4_TILE", "256259"))
NOBIAS = os.environ.get("G4_NOBIAS") == "1"
for (M, N, K) in [(203, 320, 192), (256, 256, 64), (256, 256, 128), (256, 256, 192), (256, 256, 512), (333, 576, 448), (4000, 1024, 768)]:
    for tname, epi in [("f16", 0), ("q4_1", 0), ("q4_0", 4), ("q8_0", 1)]:
        tid = ref.GGML_TYPES[tname]
        W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
        raw = ref.quantize(tid, W)
        X = rng.standard_normal((M, K)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32); resid = rng.standard_normal((M, N)).astype(np.float32)
        if NOBIAS:
            bias = None
            if epi != 4: resid = None
        base = run(tid, raw, N, K, X, bias, resid, epi, 1064064)
        nbad = 0; info = ""
        for rep in range(8):
            y = run(tid, raw, N, K, X, bias, resid, epi, TILE)
            bad = np.argwhere(~((y == base) | (np.isnan(y) & np.isnan(base))))
            if len(bad):
                nbad += 1
                if not info:
                    r, c = bad[:, 0], bad[:, 1]
                    info = "rep %d: %d elems rows %d..%d cols %d..%d (rows%%16 %s, cols%%16 %s, cols//16 %s, rows//16 %s) maxdiff %g nan %d first %s got %g want %g" % (
                        rep, len(bad), r.min(), r.max(), c.min(), c.max(), sorted(set(r % 16))[:8], sorted(set(c % 16))[:8], sorted(set(c // 16))[:12], sorted(set(r // 16))[:12],
                        np.nanmax(np.abs(y - base)), int(np.isnan(y).sum()), bad[0], y[tuple(bad[0])], base[tuple(bad[0])])
        print("%-18s %-5s epi %d: %d/8 runs differ %s" % ((M, N, K), tname, epi, nbad, info), flush=True)
