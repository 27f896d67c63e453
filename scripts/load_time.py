"""clip_model_load wall time per model (GGUF parse + host repack + H2D).  usage: python scripts/load_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import clip_cpp_amd as cc  # noqa: E402
from oracle import fixtures  # noqa: E402

for cfg, ft, kw in [("b32", "q4_0", {}), ("b32", "f16", {}), ("l14", "q5_1", dict(text=False)), ("l14", "f16", dict(text=False)), ("h14", "q8_0", dict(text=False))]:
    p = fixtures.cached_model("/tmp/clip_amd_fixtures", cfg, ft, **kw)
    open(p, "rb").read()   # page cache warm
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        c = cc.Clip(p, device=0)
        ts.append(time.perf_counter() - t0)
        c.close()
    print("%-4s %-5s %7.1f MB  load %s ms" % (cfg, ft, os.path.getsize(p) / 1e6, " ".join("%.0f" % (t * 1e3) for t in ts)), flush=True)
