"""Small-batch serving: R independent requests of B images in flight at once (R contexts of the same file on R HIP streams).
A batch-32 forward only fills part of the chip per kernel (180-450 workgroups), so concurrent requests raise the aggregate rate.
usage: python scripts/concurrent_requests.py [B] [model] [ftype]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import clip_cpp_amd as cc  # noqa: E402
from clip_cpp_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = sys.argv[2] if len(sys.argv) > 2 else "b32"
ftype = sys.argv[3] if len(sys.argv) > 3 else "q4_0"
path = synth.cached_model("/tmp/clip_amd_fixtures", model, ftype, text=False, vision=True)
for R in (1, 2, 3, 4, 6, 8):
    ctxs = []
    for r in range(R):
        c = cc.Clip(path, device=0)
        s = torch.cuda.Stream()
        c.set_stream(s.cuda_stream)
        S, proj = c.vision_config["image_size"], c.vision_config["projection_dim"]
        x = torch.randn((B, S, S, 3), dtype=torch.float32, device="cuda")
        y = torch.empty((B, proj), dtype=torch.float32, device="cuda")
        ctxs.append((c, s, x, y))

    def step():
        for c, s, x, y in ctxs:
            c.encode_images_device(x.data_ptr(), B, y.data_ptr(), True)

    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        step()
        torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s %s batch %d, %d requests in flight: %.0f img/s aggregate, %.3f ms per round" % (model, ftype, B, R, R * B * n / dt, dt / n * 1e3), flush=True)
    for c, s, x, y in ctxs:
        c.close()
