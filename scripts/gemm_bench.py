"""GEMM micro-benchmark over shapes / tiles (kernel A/B work).  usage: python scripts/gemm_bench.py [tile ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    import torch  # noqa: F401  (HIP runtime order, see tests/conftest.py)
except Exception:
    pass
import clip_cpp_amd  # noqa: E402

L = clip_cpp_amd.lib()
TYPES = {"f16": 1, "q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8}
SHAPES = [  # (name, M, N, K, epi)
    ("b32.qkv", 12800, 2304, 768, 1), ("b32.out", 12800, 768, 768, 4), ("b32.up", 12800, 3072, 768, 3), ("b32.down", 12800, 768, 3072, 4),
    ("b32.b32.up", 1600, 3072, 768, 3), ("b32.b32.down", 1600, 768, 3072, 4),
    ("l14.up", 65792, 4096, 1024, 3), ("l14.down", 65792, 1024, 4096, 4),
    ("b1.qkv", 50, 2304, 768, 1), ("b1.out", 50, 768, 768, 4), ("b1.up", 50, 3072, 768, 3), ("b1.down", 50, 768, 3072, 4),
    ("b32.b32.qkv", 1600, 2304, 768, 1), ("b32.b32.out", 1600, 768, 768, 4),
    ("txt.qkv", 10290, 1536, 512, 1), ("txt.out", 10290, 512, 512, 4), ("txt.up", 10290, 2048, 512, 3), ("txt.down", 10290, 512, 2048, 4),
    ("l14.qkv", 65792, 3072, 1024, 1), ("l14.out", 65792, 1024, 1024, 4),
    ("l14.b32.qkv", 8224, 3072, 1024, 1), ("l14.b32.out", 8224, 1024, 1024, 4), ("l14.b32.up", 8224, 4096, 1024, 3), ("l14.b32.down", 8224, 1024, 4096, 4),
    ("b64.qkv", 3200, 2304, 768, 1), ("b64.out", 3200, 768, 768, 4), ("b64.up", 3200, 3072, 768, 3), ("b64.down", 3200, 768, 3072, 4),
    ("b1024.qkv", 51200, 2304, 768, 1), ("b1024.out", 51200, 768, 768, 4), ("b1024.up", 51200, 3072, 768, 3), ("b1024.down", 51200, 768, 3072, 4),
    ("l14.b1.qkv", 257, 3072, 1024, 1), ("l14.b1.out", 257, 1024, 1024, 4), ("l14.b1.up", 257, 4096, 1024, 3), ("l14.b1.down", 257, 1024, 4096, 4),
    ("l14.b128.qkv", 32896, 3072, 1024, 1), ("l14.b128.out", 32896, 1024, 1024, 4), ("l14.b128.up", 32896, 4096, 1024, 3), ("l14.b128.down", 32896, 1024, 4096, 4),
    ("sq.k1k", 4096, 4096, 1024, 1), ("sq.k8k", 4096, 4096, 8192, 1), ("sq8.k4k", 8192, 8192, 4096, 1),
    ("b128.qkv", 6400, 2304, 768, 1), ("b128.out", 6400, 768, 768, 4), ("b128.up", 6400, 3072, 768, 3), ("b128.down", 6400, 768, 3072, 4),
]
for a in sys.argv[1:]:      # custom shapes: MxNxK[:epi]  (epi as clip_amd_bench_gemm: 1 f16, 3 quick-gelu f16, 4 residual)
    if a.count("x") == 2 and a.replace("x", "").replace(":", "").isdigit():
        dims, _, e = a.partition(":")
        M_, N_, K_ = (int(v) for v in dims.split("x"))
        SHAPES.append((a + ".", M_, N_, K_, int(e) if e else 1))
tiles = [int(t) for t in sys.argv[1:] if t.isdigit()] or [0]
if 'ksweep' in sys.argv[1:]:
    tiles = [0] + [ks * 1000000 + t for t in (64064, 64128, 128128, 160128) for ks in (1, 2, 3, 4, 6, 8)]
types = [t for t in sys.argv[1:] if t in TYPES] or ["q4_0"]
only = [a for a in sys.argv[1:] if "." in a] + [a + "." for a in sys.argv[1:] if a.count("x") == 2]
debug = [int(a[3:]) for a in sys.argv[1:] if a.startswith("dbg")] or [0]
ITERS = int(os.environ.get("GEMM_ITERS", "20"))   # long runs (thousands) show the sustained, power-limited rate
PRE = (1 << 16) if "pre" in sys.argv[1:] else 0   # 8-wave kernel: time the GEMM alone on an already dequantised fp16 panel (per-layer form)
FOLD = (1 << 17) if "fold" in sys.argv[1:] else 0  # epilogues in their LayerNorm-fold form (consumer: fp16 epilogues; producer: residual epilogue)
if "blas" in sys.argv[1:]:
    # yardstick: the vendor library (hipBLASLt / rocBLAS through torch) on the same shapes, plain f16 x f16 -> f16, no epilogue
    for name, M, N, K, epi in SHAPES:
        if only and name not in only:
            continue
        x = torch.randn(M, K, device="cuda", dtype=torch.float16)
        w = torch.randn(N, K, device="cuda", dtype=torch.float16)
        for _ in range(3):
            y = torch.nn.functional.linear(x, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            y = torch.nn.functional.linear(x, w)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print("blas  %-14s M=%6d N=%5d K=%5d | %8.1f us %7.1f TF" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
    sys.exit(0)
for tname in types:
    for name, M, N, K, epi in SHAPES:
        if only and name not in only:
            continue
        row = []
        for tile in tiles:
            for dbg in debug:
                us = L.clip_amd_bench_gemm(TYPES[tname], N, K, M, epi | (dbg << 8) | PRE | FOLD, tile, ITERS)
                row.append("%7d%s: %8.1f us %7.1f TF" % (tile, ("/d%d" % dbg) if dbg else "", us, 2.0 * M * N * K / us / 1e6 if us > 0 else -1))
        print("%-5s %-14s M=%6d N=%5d K=%5d | %s" % (tname, name, M, N, K, " | ".join(row)), flush=True)
